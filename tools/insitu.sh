#!/bin/bash
# usage: tools/insitu.sh <name> [lib.so]  -> per (kernel, grid) average duration of the launches inside the bench step
# (rocprofv3 --kernel-trace over a short bench run), written to gpurun_out/insitu_<name>.txt
name=$1; lib=$2
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/insitu_$name
mkdir -p $out
cd $root
[ -n "$lib" ] && export CENTERCLIP_HIP_LIB=$lib
rocprofv3 --kernel-trace -d $out -o t --output-format csv -- python bench.py --steps 10 --warmup 2 --min-seconds 0.2 --no-cpu-baseline --no-extras > $out/run.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $root/gpurun_out/insitu_$name.txt
import csv, sys, collections, re
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    key = (n[:60], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(key, [0, 0])
    a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%-60s g=%-8s,%-4s b=%-4s n=%-6d avg=%8.2f us  %5.1f%%" % (k[0], k[1], k[2], k[3], v[0], v[1] / v[0] / 1e3, 100.0 * v[1] / tot))
PY
