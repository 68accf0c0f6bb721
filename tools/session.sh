#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out
tag=${1:-r4u}
timeout 900 python -m pytest tests/test_clip_gpu.py tests/test_r2_gpu.py tests/test_r3_gpu.py tests/test_r4_gpu.py -x -q > $o/${tag}_tests.txt 2>&1; tail -3 $o/${tag}_tests.txt
for i in 1 2 3; do for v in nopf pf pf2; do echo -n "$v "; CENTERCLIP_HIP_LIB=$PWD/ab/lib_$v.so python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; done; done | paste - - - - - - - - -
for v in wallnopf wall; do echo $v; CENTERCLIP_HIP_LIB=$PWD/ab/lib_$v.so python tools/gemm_timeline.py 2>&1 | head -2; done
