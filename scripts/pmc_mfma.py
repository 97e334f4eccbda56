"""Aggregate one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU
SQ_WAVE_CYCLES GRBM_GUI_ACTIVE) per kernel name (JSON to stdout).

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x 2.4 GHz x kernel duration): the fraction of the chip's MFMA issue
capacity at the peak clock that the kernel used (SQ_VALU_MFMA_BUSY_CYCLES counts 32 cycles per v_mfma_f32_32x32x16_bf16
and 16 per 16x16x32, summed over every SIMD: /opt/skills/guides/MI355X_MICROARCH.md constants table) -- directly
comparable with `roofline.frac` (FLOPs / 2.5 PF).  mfma_busy_of_active divides by the cycles the GPU was actually
active (GRBM_GUI_ACTIVE is summed over the 8 XCDs) instead: it removes the clock the profiled run ran at.
Durations come from the dispatch timestamps of the same pass (counter collection serialises dispatches and lowers the
clock, so they are longer than in the un-profiled run; ratios between kernels hold)."""
import collections
import csv
import datetime
import glob
import json
import sys

d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        try:
            dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
        except (KeyError, ValueError):
            pass
seen = set()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (name, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            cnt[name] += 1
            t = dur.get(r["Dispatch_Id"])
            if t is None and r.get("End_Timestamp") and r.get("Start_Timestamp"):
                t = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
            acc[name]["_seconds"] += t or 0.0
out = {}
for name, c in sorted(acc.items()):
    n = max(cnt[name], 1)
    sec = c["_seconds"]
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    out[name] = {
        "launches": cnt[name], "avg_us_profiled": sec / n * 1e6,
        "mfma_busy_cycles_per_launch": mf / n,
        "mfma_busy": mf / (1024.0 * 2.4e9 * sec) if sec > 0 else None,
        "mfma_busy_of_active": mf / (1024.0 * gui / 8.0) if gui > 0 else None,
        "sq_busy_cycles_per_launch": c.get("SQ_BUSY_CYCLES", 0.0) / n,
        "sq_wait_inst_lds_per_wave_cycle": (c.get("SQ_WAIT_INST_LDS", 0.0) / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None,
        "valu_insts_per_launch": c.get("SQ_INSTS_VALU", 0.0) / n,
        "effective_clock_ghz": (gui / 8.0 / sec * 1e-9) if sec > 0 and gui > 0 else None,
    }
try:
    commit = open(".head_commit").read().strip()
except OSError:
    commit = None
print(json.dumps({"commit": commit, "date": datetime.datetime.utcnow().strftime("%Y-%m-%d"),
                  "note": "one rocprofv3 --kernel-trace --pmc pass over bench.py --steps 2 --warmup 2 --no-graph; "
                          "mfma_busy = MFMA busy cycles / (1024 SIMDs x 2.4 GHz x dispatch duration)",
                  "kernels": out}, indent=1))
