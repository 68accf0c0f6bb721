#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s12
timeout 900 python -m pytest tests/test_r4_gpu.py -x -q -m gpu -k "inproj_attention" 2>&1 | tail -3
for v in wall3g wall3; do echo "== $v"; CENTERCLIP_HIP_LIB=$PWD/ab/lib_$v.so python tools/fused_timeline.py long 2>&1 | grep -v amdgpu.ids; done > gpurun_out/s12/timeline.txt 2>&1
cat gpurun_out/s12/timeline.txt
for i in 1 2 3; do
  for v in guarded new; do
    echo -n "$v cfg5 "
    CENTERCLIP_HIP_LIB=$PWD/ab/lib_$v.so python bench.py --workload cfg5 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done > gpurun_out/s12/ab.txt 2>&1
cat gpurun_out/s12/ab.txt
