#!/bin/bash
# LDS bank conflicts and MFMA busy cycles of the weight-gradient kernel (tools/native/wgrad_probe; counters + kernel trace only,
# one counter group per run)
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_wgrad
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d $out/p1 -o w --output-format csv -- tools/native/wgrad_probe > $out/p1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace -d $out/p2 -o w --output-format csv -- tools/native/wgrad_probe > $out/p2.log 2>&1
python - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_wgrad")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad_tn" in r["Kernel_Name"]:
            acc[(r["Grid_Size"] if "Grid_Size" in r else "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g, d in sorted(acc.items(), key=lambda kv: -len(kv[1])):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    util = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0) if gui else None
    conf = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0)
    print("grid", g, {k: round(v, 1) for k, v in m.items()}, "| bank-conflict cycles / LDS active cycles %.4f" % conf, "| mfma_util", None if util is None else round(util, 3))
PY
