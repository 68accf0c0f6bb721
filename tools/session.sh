#!/bin/bash
# round 5, session 6: tile 12 (128x192, 4 waves, two workgroups per CU) for the fp16-output LN-folded GEMMs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s6
python tools/tile_sweep.py 9600 2400 12800 25600 37824 6464 > gpurun_out/s6/tile_sweep.txt 2>&1
cat gpurun_out/s6/tile_sweep.txt
export CENTERCLIP_HIP_LIB=$PWD/ab/lib_dev.so
for i in 1 2 3; do
  for v in "" "CC_TILE_E6_B=12" "CC_TILE_E6_B=12 CC_TILE_E6_S=12"; do
    echo -n "[$v] cfg2 "
    env $v python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done > gpurun_out/s6/ab.txt 2>&1
cat gpurun_out/s6/ab.txt
