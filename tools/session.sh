#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_r4_gpu.py tests/test_r3_gpu.py tests/test_r2_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
  for v in "" "$PWD/ab/lib_not10.so"; do
    for k in cfg2 cfg4 cfg5 cfg3; do
      echo -n "[$v] $k "
      if [ -n "$v" ]; then export CENTERCLIP_HIP_LIB=$v; else unset CENTERCLIP_HIP_LIB; fi
      python bench.py --workload $k --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
    done
  done
done
