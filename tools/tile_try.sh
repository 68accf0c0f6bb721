#!/bin/bash
# step-level A/B of tile overrides inside one session: tools/tile_try.sh VAR=val [VAR=val ...]  (each tried alone)
run() { python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2; do
echo "baseline $(run)"
for v in "$@"; do echo "$v $(env $v bash -c "$(declare -f run); run")"; done
done
