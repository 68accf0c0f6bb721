cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_clustered_r06
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace -d $out/g -o g --output-format csv -- python tools/kernels_clustered.py > $out/g.log 2>&1
python - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_clustered_r06")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "g", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, grid), d in sorted(acc.items()):
    if "gemm_f16" not in k: continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    util = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024) if gui else 0
    print("%-60s grid %-8s n=%d  MFMA_BUSY %.3g  GUI_ACTIVE %.3g  mfma_util %.1f %%" % (k[:60], grid, len(d["GRBM_GUI_ACTIVE"]), m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), gui, 100 * util))
PY
