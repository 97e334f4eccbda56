"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel name (JSON to stdout).
Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters
are reported in KB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at
64 bytes, so it is doubled; WRITE_SIZE matched the exact output size of the GEMMs and is taken
as is (profiles/r01_pmc_gemm.txt)."""
import csv, glob, json, sys, collections


def load(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0, collections.Counter()])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            a = acc[name]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            a[2][float(r["Counter_Value"])] += 1          # per-dispatch values (summed over the XCDs by rocprofv3)
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for name in sorted(set(fetch) | set(write)):
    fk, fn, fh = fetch.get(name, [0.0, 0, {}])
    wk, wn, _ = write.get(name, [0.0, 0, {}])
    out[name] = {"launches": max(fn, wn),
                 "fetch_bytes_per_launch": 2.0 * 1024.0 * fk / max(fn, 1),
                 "write_bytes_per_launch": 1024.0 * wk / max(wn, 1)}
    if "--clusters" in sys.argv and fn > 1:
        # a kernel name covers several problem shapes (the 64x64 chain: K = 512 / 1536 / 2048): launches binned by MB fetched
        bins = collections.Counter()
        for v, c in fh.items():
            bins[int(round(2.0 * 1024.0 * v / 2e6)) * 2] += c
        out[name]["fetch_mb_bins"] = {str(k): bins[k] for k in sorted(bins)}
import datetime
if "--decode" in sys.argv:
    # the decode leg: everything the job launched, per decode step (k_beam_prepare runs once per step)
    steps = max(out.get("k_beam_prepare", {}).get("launches", 0), 1)
    tot_f = sum(v["fetch_bytes_per_launch"] * v["launches"] for v in out.values())
    tot_w = sum(v["write_bytes_per_launch"] * v["launches"] for v in out.values())
    try:
        commit = open(".head_commit").read().strip()
    except OSError:
        commit = None
    print(json.dumps({"commit": commit, "date": datetime.datetime.utcnow().strftime("%Y-%m-%d"),
                      "note": "FETCH_SIZE KB x2 (gfx950 correction) + WRITE_SIZE KB x1 of EVERY kernel of bench.py --mode decode "
                              "--sentences 320 --decode-streams 1 (encoder passes and start-up included), divided by the decode steps",
                      "decode_steps": steps, "fetch_bytes_per_step": tot_f / steps, "write_bytes_per_step": tot_w / steps,
                      "bytes_per_step": (tot_f + tot_w) / steps,
                      "by_kernel_bytes_per_step": {k: (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"] / steps
                                                   for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_launch"] + kv[1]["write_bytes_per_launch"]) * kv[1]["launches"])[:12]}},
                     indent=1))
    sys.exit(0)
try:
    commit = open(".head_commit").read().strip()   # written by scripts/gpu.sh before the snapshot travels
except OSError:
    commit = None
print(json.dumps({"commit": commit, "date": datetime.datetime.utcnow().strftime("%Y-%m-%d"), "note": "FETCH_SIZE KB x2 (gfx950 correction), WRITE_SIZE KB x1; per-launch averages over "
                          "bench.py --steps 2 --warmup 2 --no-graph", "kernels": out}, indent=1))
