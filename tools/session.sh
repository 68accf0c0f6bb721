#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out
tag=${1:-r4p}
bash tools/ab.sh patch1 patch6 3 > $o/${tag}_ab.txt 2>&1; cat $o/${tag}_ab.txt | paste - - - - - -
echo "dev: default vs tile 10 / 6 for the clustered LN GEMMs"
for i in 1 2 3; do
  for env in "X=1" "CC_TILE_E5_S=10 CC_TILE_E6_S=10" "CC_TILE_E5_S=10" "CC_TILE_E6_S=10" "CC_TILE_E5_S=6 CC_TILE_E6_S=6"; do
    echo -n "$env: "; env $env CENTERCLIP_HIP_LIB=$PWD/ab/lib_dev.so python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
bash tools/insitu.sh ${tag}_patch6 $PWD/ab/lib_patch6.so > /dev/null 2>&1; head -12 $o/insitu_${tag}_patch6.txt
timeout 900 python -m pytest tests/test_clip_gpu.py tests/test_r3_gpu.py -x -q > $o/${tag}_tests.txt 2>&1; tail -3 $o/${tag}_tests.txt
