#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s14
timeout 1200 python -m pytest tests/test_r4_gpu.py tests/test_clip_gpu.py tests/test_r2_gpu.py tests/test_r3_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do
  for v in single new; do
    for k in cfg2 cfg3 cfg4; do
      echo -n "$v $k "
      CENTERCLIP_HIP_LIB=$PWD/ab/lib_$v.so python bench.py --workload $k --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
    done
  done
done > gpurun_out/s14/ab.txt 2>&1
cat gpurun_out/s14/ab.txt
