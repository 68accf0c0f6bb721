#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out
tag=${1:-r4e}
timeout 900 python -m pytest tests/test_r4_gpu.py tests/test_clip_gpu.py -x -q > $o/${tag}_tests.txt 2>&1; tail -3 $o/${tag}_tests.txt
CENTERCLIP_HIP_LIB=$PWD/ab/lib_wall.so timeout 600 python tools/gemm_timeline.py > $o/${tag}_timeline.txt 2>&1; cat $o/${tag}_timeline.txt | cut -c1-400
timeout 600 python tools/resid_sweep.py > $o/${tag}_resid.txt 2>&1; cat $o/${tag}_resid.txt | cut -c1-250
bash tools/ab.sh nostag stag 3 > $o/${tag}_ab_nostag_stag.txt 2>&1; cat $o/${tag}_ab_nostag_stag.txt | paste - - - - - -
bash tools/ab.sh stag stagsc1 3 > $o/${tag}_ab_stag_sc1.txt 2>&1; cat $o/${tag}_ab_stag_sc1.txt | paste - - - - - -
for v in nostag stag; do bash tools/insitu.sh ${tag}_$v $PWD/ab/lib_$v.so > /dev/null 2>&1; echo "== $v"; head -5 $o/insitu_${tag}_$v.txt; done
timeout 1500 python -m pytest tests/test_r3_gpu.py tests/test_r2_gpu.py -x -q > $o/${tag}_tests_clip.txt 2>&1; tail -3 $o/${tag}_tests_clip.txt
