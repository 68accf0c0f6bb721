#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out
tag=${1:-r4q}
timeout 600 python -m pytest tests/test_r4_gpu.py -x -q -k "inproj" > $o/${tag}_tests_new.txt 2>&1; tail -15 $o/${tag}_tests_new.txt
timeout 900 python -m pytest tests/test_clip_gpu.py tests/test_r3_gpu.py -x -q > $o/${tag}_tests.txt 2>&1; tail -5 $o/${tag}_tests.txt
bash tools/ab.sh nofuse fuse 3 > $o/${tag}_ab.txt 2>&1; cat $o/${tag}_ab.txt | paste - - - - - -
bash tools/insitu.sh ${tag}_fuse $PWD/ab/lib_fuse.so > /dev/null 2>&1; head -14 $o/insitu_${tag}_fuse.txt
