#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out
tag=${1:-r4j}
timeout 2400 python -m pytest tests -x -q -m gpu > $o/${tag}_tests_all.txt 2>&1; tail -4 $o/${tag}_tests_all.txt
python bench.py --steps 30 --warmup 3 > $o/${tag}_bench.json 2> $o/${tag}_bench.err; tail -c 1500 $o/${tag}_bench.json; echo; tail -3 $o/${tag}_bench.err
