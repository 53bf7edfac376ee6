"""Averages rocprofv3 --pmc CSVs per kernel and counter: python tools/summarise_pmc.py <dir with one sub-directory per pass>.
Prints one JSON object {kernel: {counter: mean per dispatch, ..., "dispatches": n}} (kernel names shortened)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            if "fbsmi" not in k:
                continue
            k = k.replace("(anonymous namespace)::", "").replace("void ", "").replace("fbsmi::", "")
            k = re.sub(r"\(.*", "", k) + " grid=" + r["Grid_Size"]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[k]["dispatches"] = max(len(v) for v in cs.values())
print(json.dumps(out, indent=1))
