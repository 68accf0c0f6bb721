#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out
tag=${1:-r4k}
timeout 2400 python -m pytest tests -q -m gpu > $o/${tag}_tests_all.txt 2>&1; tail -12 $o/${tag}_tests_all.txt; grep -n "relative errors" $o/${tag}_tests_all.txt | head
