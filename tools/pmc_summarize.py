"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports exactly half of the bytes of a wide
coalesced streaming read -> doubled here; values are in KiB-ish units of the tool (x1024 -> bytes)."""
import csv, glob, json, os, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for which, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(out, which, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == key:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][which + "_raw_per_launch"] = sum(v) / len(v)
        res[k]["launches"] = len(v)
OURS = ("gemm_f16_kernel", "kmedoids_select", "gram_dist", "lp_dist", "reduce_tokens", "token_norm", "attention_",
        "im2col", "sim_prepare", "row_stats", "layernorm_kernel", "head_project")
summary = {}
for k, v in res.items():
    if not any(s in k for s in OURS):        # torch's own fill / randn kernels of the driver script are not ours
        continue
    f = v.get("fetch_raw_per_launch", 0.0) * 1024.0 * 2.0       # x2: gfx950 FETCH_SIZE under-count
    w = v.get("write_raw_per_launch", 0.0) * 1024.0
    summary[k] = dict(launches=v.get("launches", 0), fetch_bytes_per_launch=f, write_bytes_per_launch=w,
                      hbm_bytes_per_launch=f + w)
json.dump(summary, open(os.path.join(out, "traffic.json"), "w"), indent=1)
for k, v in sorted(summary.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:14]:
    print("%-64s n=%3d fetch %9.2f MB  write %9.2f MB" % (k[:64], v["launches"], v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))
