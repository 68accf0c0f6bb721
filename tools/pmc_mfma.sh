#!/bin/bash
# MFMA utilisation counters for the kernels launched by tools/kernels_for_pmc.py (own pass, counters + kernel trace only)
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_mfma
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace -d $out/p1 -o m --output-format csv -- python tools/kernels_for_pmc.py > $out/p1.log 2>&1
python - <<'PY'
import csv, glob, collections, os, json
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_mfma")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p1", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in acc.items():
    if not any(s in k for s in ("gemm_f16", "gram_dist", "kmedoids", "sim_prepare")):
        continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    mfma = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    m["mfma_util"] = mfma / (gui / 8.0 * 1024.0) if gui else None   # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    res[k] = m
    print(k[:60], {a: round(b, 4) if isinstance(b, float) else b for a, b in m.items()})
json.dump(res, open(os.path.join(out, "mfma_util.json"), "w"), indent=1)
PY
