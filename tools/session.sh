#!/bin/bash
# one GPU session of round 4 (dev tool): tests of the new kernels, stand-alone sweeps, A/B of library builds, in-situ traces
cd $GRAFT_REPO_ROOT
o=gpurun_out
tag=${1:-r4b}
timeout 900 python -m pytest tests/test_r4_gpu.py tests/test_clip_gpu.py -x -q > $o/${tag}_tests.txt 2>&1; tail -3 $o/${tag}_tests.txt
timeout 600 python tools/resid_sweep.py > $o/${tag}_resid.txt 2>&1; cat $o/${tag}_resid.txt | cut -c1-250
timeout 600 python tools/tile_sweep.py > $o/${tag}_tiles.txt 2>&1; cat $o/${tag}_tiles.txt | cut -c1-250
for pair in "noinit init" "init initnt" "init initsc1"; do set -- $pair; bash tools/ab.sh $1 $2 2 > $o/${tag}_ab_$1_$2.txt 2>&1; cat $o/${tag}_ab_$1_$2.txt | paste - - - - ; done
for v in noinit init initsc1; do bash tools/insitu.sh ${tag}_$v $PWD/ab/lib_$v.so > /dev/null 2>&1; echo "== $v"; head -9 $o/insitu_${tag}_$v.txt; done
timeout 1500 python -m pytest tests/test_r3_gpu.py tests/test_r2_gpu.py tests/test_spectral_gpu.py -x -q > $o/${tag}_tests_clip.txt 2>&1; tail -3 $o/${tag}_tests_clip.txt
