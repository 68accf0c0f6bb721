#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print('quick ms_per_step', json.loads(sys.stdin.read())['ms_per_step'])"
bash tools/refresh_profiles.sh r05 2>&1 | tail -14
