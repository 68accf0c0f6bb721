#!/bin/bash
# A/B two builds of the library inside ONE gpurun session (box-to-box spread is larger than most kernel changes):
#   build variant A, cp centerclip_amd/lib/libcenterclip_hip.so ab/lib_a.so; build B -> ab/lib_b.so; then
#   gpurun -- 'bash tools/ab.sh a b [rounds]'      (ab/ is git-ignored but travels with the snapshot)
a=$1; b=$2; n=${3:-3}
for i in $(seq $n); do
  for v in $a $b; do
    echo -n "$v "
    CENTERCLIP_HIP_LIB=$PWD/ab/lib_$v.so python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 |
      python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
