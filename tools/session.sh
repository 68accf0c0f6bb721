#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_clip_gpu.py tests/test_cfg1_plumbing.py -x -q -m gpu 2>&1 | tail -3
