#!/bin/bash
# round 5, session 7: the round's committed profiles (PMC traffic + MFMA utilisation stand-alone, bench line, kernel stats of the
# same bench command) + the parity tests of cfg 3 / cfg 5 at their full per-GPU batch
cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh r05 2>&1 | tail -40
( time timeout 1500 python -m pytest tests/test_r3_gpu.py -x -q -m gpu -k paired_towers -s ) > gpurun_out/profiles/pytest_paired.txt 2>&1
grep "normalised\|passed\|failed\|real" gpurun_out/profiles/pytest_paired.txt
