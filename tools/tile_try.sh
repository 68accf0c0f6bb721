#!/bin/bash
# step-level A/B of environment switches inside one session: tools/tile_try.sh [-n rounds] VAR=val [VAR=val ...]  (each tried alone)
n=2; if [ "$1" = "-n" ]; then n=$2; shift 2; fi
run() { python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in $(seq $n); do
echo "baseline $(run)"
for v in "$@"; do echo "$v $(env $v bash -c "$(declare -f run); run")"; done
done
