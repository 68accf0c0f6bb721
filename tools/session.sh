#!/bin/bash
# round 5, session 5: fused attention tile skipping the fragment rows behind its last sequence (A/B), tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5
timeout 900 python -m pytest tests/test_r4_gpu.py tests/test_r3_gpu.py -x -q -m gpu -k "inproj_attention or paired_towers or timed_step or step" > gpurun_out/s5/pytest.txt 2>&1
tail -4 gpurun_out/s5/pytest.txt
for i in 1 2 3; do
  for v in allrows new; do
    for k in cfg5 cfg4 cfg2; do
      echo -n "$v $k "
      CENTERCLIP_HIP_LIB=$PWD/ab/lib_$v.so python bench.py --workload $k --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 |
        python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
    done
  done
done > gpurun_out/s5/ab.txt 2>&1
cat gpurun_out/s5/ab.txt
