#!/bin/bash
# FETCH_SIZE per access pattern on a known byte count (MI355X_MICROARCH.md: only the wide coalesced read is calibrated).
# Output: gpurun_out/fetch_calibration.txt  (copy to profiles/rNN_fetch_size_calibration.txt)
cd "$(dirname "$0")/.."; export PYTHONPATH=$PWD TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf gpurun_out/pmcc
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OLDPWD/gpurun_out/pmcc -o p --output-format csv -- \
   python $OLDPWD/scripts/fetch_calibrate.py > $OLDPWD/gpurun_out/pmcc.log 2>&1); echo "rc=$?"
python - <<'P' > gpurun_out/fetch_calibration.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "k_probe_read" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
what = {"0": "16 B / lane, 1 KB contiguous per wave instruction", "1": "MFMA-fragment pattern: 16 rows x 64 B per instruction",
        "2": "4 B / lane (256 B per instruction)", "3": "8 B / lane (512 B per instruction)", "4": "2 B / lane (128 B per instruction)"}
print("# zk_probe_read: 1 GiB (1073741824 bytes) read once per launch; FETCH_SIZE as reported (KB), bytes / (FETCH_SIZE x 1024)")
for k in sorted(acc):
    pat = k.split("<")[1].split(">")[0]
    for v in acc[k]:
        print("%-22s %-52s FETCH_SIZE %12.0f KB   true / reported = %.3f" % (k, what.get(pat, ""), v, (1 << 30) / (v * 1024.0)))
P
rm -rf gpurun_out/pmcc; cat gpurun_out/fetch_calibration.txt
