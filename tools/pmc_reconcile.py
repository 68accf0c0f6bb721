"""Per-kernel averages of every counter collected by tools/pmc_reconcile.sh -> <out>/counters.json + a table."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "g*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"] + " grid " + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
KEEP = ("gemm_f16_kernel", "kmedoids_select", "gram_dist", "attention", "elementwise", "copy", "Memcpy", "im2col", "vectorized")
res = {}
for k, cs in acc.items():
    if not any(s in k for s in KEEP):
        continue
    res[k] = {c: dict(avg=sum(v) / len(v), n=len(v), last=v[-1]) for c, v in cs.items()}
json.dump(res, open(os.path.join(out, "counters.json"), "w"), indent=1)
names = sorted({c for v in res.values() for c in v})
for k, v in sorted(res.items()):
    print("==", k[:110])
    for c in names:
        if c in v:
            print("    %-36s avg %16.1f  last %16.1f  n=%d" % (c, v[c]["avg"], v[c]["last"], v[c]["n"]))
