#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out
tag=${1:-r4l}
bash tools/ab.sh plainf16 wtgelu 3 > $o/${tag}_ab.txt 2>&1; cat $o/${tag}_ab.txt | paste - - - - - -
for v in plainf16 wtgelu; do bash tools/insitu.sh ${tag}_$v $PWD/ab/lib_$v.so > /dev/null 2>&1; echo "== $v"; head -4 $o/insitu_${tag}_$v.txt; done
bash tools/refresh_profiles.sh r04 2>&1 | tail -40
bash tools/pmc_insitu.sh > $o/${tag}_pmc_insitu.txt 2>&1; tail -16 $o/${tag}_pmc_insitu.txt
