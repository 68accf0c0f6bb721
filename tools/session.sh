#!/bin/bash
# the last GPU session's command list (rewritten per session)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
bash tools/refresh_profiles.sh r04 > gpurun_out/refresh.log 2>&1
bash tools/pmc_insitu.sh > gpurun_out/pmc_insitu.log 2>&1
