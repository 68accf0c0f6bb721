#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out
for i in 1 2 3; do for v in base stg stgwt; do echo -n "$v "; CENTERCLIP_HIP_LIB=$PWD/ab/lib_$v.so python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; done; done | paste - - - - - - - - -
for v in stg stgwt; do CENTERCLIP_HIP_LIB=$PWD/ab/lib_$v.so timeout 300 python -m pytest tests/test_r4_gpu.py -x -q -k inproj 2>&1 | tail -1; done
