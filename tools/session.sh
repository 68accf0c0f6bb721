#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_r5_gpu.py tests/test_r4_gpu.py -m gpu -x -q 2>&1 | tail -3
bash tools/train_prof.sh 8 2>&1 | cut -c1-170 | tail -10
python examples/train_synthetic.py --steps 1 2>&1 | tail -1
for i in 1 2 3; do
  for v in "" "$PWD/ab/lib_nors.so"; do
    if [ -n "$v" ]; then export CENTERCLIP_HIP_LIB=$v; else unset CENTERCLIP_HIP_LIB; fi
    echo -n "[$v] cfg2 "
    python bench.py --steps 30 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
