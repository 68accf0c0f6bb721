#!/bin/bash
# usage: tools/prof.sh <name> <cmd...>   -> gpurun_out/prof_<name>/ kernel stats (rocprofv3 --kernel-trace --stats)
name=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$name
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $out -o $name --output-format csv -- "$@" > $out/run.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
echo "== $f"
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-70s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for r in rows[:25]:
    print("%-70s %8s %12.1f %10.2f %6.1f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
