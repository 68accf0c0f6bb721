#!/bin/bash
# try tile overrides for the small-M (clustered) blocks; prints ms/step
run() { python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "baseline $(run)"
for e in 5 6 7; do for t in 1 2 3 4; do echo "E${e}_S=$t $(env CC_TILE_E${e}_S=$t bash -c "$(declare -f run); run")"; done; done
for e in 5 6 7; do for t in 1 5 6; do echo "E${e}_B=$t $(env CC_TILE_E${e}_B=$t bash -c "$(declare -f run); run")"; done; done
