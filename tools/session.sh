#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_r5_gpu.py tests/test_r4_gpu.py -m gpu -x -q 2>&1 | tail -6
bash tools/train_prof.sh 30 2>&1 | cut -c1-170
