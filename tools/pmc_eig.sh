#!/bin/bash
# Instruction / LDS counters of the spectral eigensolver kernels (own passes, counters + kernel trace only) -> gpurun_out/pmc_eig/summary.txt
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_eig
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $out/p1 -o e --output-format csv -- python tools/eig_prof.py 48 196 49 > $out/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --kernel-trace -d $out/p2 -o e --output-format csv -- python tools/eig_prof.py 48 196 49 > $out/p2.log 2>&1
python - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_eig")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for k, d in acc.items():
        if "sym_eig" not in k:
            continue
        line = k[:70] + "  " + "  ".join(f"{c}={sum(v) / len(v):.3g}" for c, v in sorted(d.items()))
        print(line); fh.write(line + "\n")
PY
