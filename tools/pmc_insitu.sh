#!/bin/bash
# HBM-traffic counters of the kernels INSIDE the bench step (same counter passes as tools/pmc.sh, on bench.py itself) ->
# gpurun_out/pmc_insitu/traffic.json, to hold against the stand-alone launches of tools/kernels_for_pmc.py: what a
# launch fetches when its operands were written by the launch before it
cd $GRAFT_REPO_ROOT
bash tools/pmc.sh insitu python bench.py --steps 4 --warmup 1 --min-seconds 0.02 --no-cpu-baseline --no-extras
