#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out
tag=${1:-r4d}
timeout 900 python -m pytest tests/test_r4_gpu.py tests/test_clip_gpu.py -x -q > $o/${tag}_tests.txt 2>&1; tail -3 $o/${tag}_tests.txt
CENTERCLIP_HIP_LIB=$PWD/ab/lib_wall.so timeout 600 python tools/gemm_timeline.py > $o/${tag}_timeline.txt 2>&1; cat $o/${tag}_timeline.txt | cut -c1-400
timeout 600 python tools/resid_sweep.py > $o/${tag}_resid.txt 2>&1; cat $o/${tag}_resid.txt | cut -c1-250
bash tools/ab.sh nopf pf 3 > $o/${tag}_ab_nopf_pf.txt 2>&1; cat $o/${tag}_ab_nopf_pf.txt | paste - - - - - -
echo "dev: default vs tile 10 for the clustered LN GEMMs"
for i in 1 2 3; do
  for env in "X=1" "CC_TILE_E5_S=10 CC_TILE_E6_S=10" "CC_TILE_E5_S=10" "CC_TILE_E6_S=10"; do
    echo -n "$env: "; env $env CENTERCLIP_HIP_LIB=$PWD/ab/lib_dev.so python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
for v in nopf pf; do bash tools/insitu.sh ${tag}_$v $PWD/ab/lib_$v.so > /dev/null 2>&1; echo "== $v"; head -9 $o/insitu_${tag}_$v.txt; done
timeout 1500 python -m pytest tests/test_r3_gpu.py tests/test_r2_gpu.py -x -q > $o/${tag}_tests_clip.txt 2>&1; tail -3 $o/${tag}_tests_clip.txt
