#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out
tag=${1:-r4g}
timeout 600 python -m pytest tests/test_r4_gpu.py -x -q > $o/${tag}_tests.txt 2>&1; tail -3 $o/${tag}_tests.txt
timeout 600 python tools/tile_sweep.py 9600 19200 > $o/${tag}_tiles.txt 2>&1; cat $o/${tag}_tiles.txt | cut -c1-250
bash tools/ab.sh nopersist persist 3 > $o/${tag}_ab.txt 2>&1; cat $o/${tag}_ab.txt | paste - - - - - -
for v in persist; do bash tools/insitu.sh ${tag}_$v $PWD/ab/lib_$v.so > /dev/null 2>&1; echo "== $v"; head -6 $o/insitu_${tag}_$v.txt; done
