#!/bin/bash
# A/B two builds of the library inside ONE gpurun session (box-to-box spread is larger than most kernel changes):
#   build variant A, cp centerclip_amd/lib/libcenterclip_hip.so abx/lib_a.so; build B -> abx/lib_b.so; then
#   gpurun -- 'bash tools/ab.sh a b [rounds]'      (abx/ is git-ignored but travels (delete it after the experiment: it rides every gpurun push) with the snapshot)
a=$1; b=$2; n=${3:-3}
for i in $(seq $n); do
  for v in $a $b; do
    echo -n "$v "
    CENTERCLIP_HIP_LIB=$PWD/abx/lib_$v.so python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras $AB_ARGS 2>/dev/null | tail -1 |
      python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
