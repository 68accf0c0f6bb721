#!/bin/bash
# Per-kernel time of the captured training step (examples/train_synthetic.py): rocprofv3 --kernel-trace over 2 eager + 14
# replayed steps, printed per step (totals / 16; the two eager steps launch the same kernels).   bash tools/train_prof.sh [rows]
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/trainprof
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o tp -- python /root/repo/examples/train_synthetic.py --steps 1 > $OUT.log 2>&1
tail -2 $OUT.log
python - "$OUT" "${1:-40}" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 1 + 2 + 1 + 3 + 10          # train_epoch's step, GraphedTrainStep's two eager ones, its first replay, the timed loop
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step %.2f ms (%d steps)" % (tot / steps / 1e6, steps))
for r in rows[:int(sys.argv[2])]:
    print("%-100s %6.1f calls/step %8.1f us avg %7.3f ms/step" % (r["Name"][:100], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                                                  float(r["TotalDurationNs"]) / steps / 1e6))
PY
