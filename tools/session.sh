#!/bin/bash
# round 5, session 2: long-form fused attention (tests + towers), raster A/B, residual tile sweep at the big-M shapes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
timeout 900 python -m pytest tests/test_r4_gpu.py -x -q -m gpu -k "inproj_attention" > gpurun_out/s2/pytest_attn.txt 2>&1
tail -5 gpurun_out/s2/pytest_attn.txt
timeout 900 python -m pytest tests/test_r3_gpu.py -x -q -m gpu -k "paired_towers" > gpurun_out/s2/pytest_towers.txt 2>&1
tail -5 gpurun_out/s2/pytest_towers.txt
for i in 1 2; do
  for v in g8 new; do
    for k in cfg2 cfg4 cfg5; do
      echo -n "$v $k "
      CENTERCLIP_HIP_LIB=$PWD/ab/lib_$v.so python bench.py --workload $k --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 |
        python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
    done
  done
done > gpurun_out/s2/ab.txt 2>&1
cat gpurun_out/s2/ab.txt
python tools/resid_sweep.py big > gpurun_out/s2/resid_sweep_big.txt 2>&1
cat gpurun_out/s2/resid_sweep_big.txt
for k in cfg5 cfg4; do
  bash tools/prof.sh fwd2_$k python bench.py --workload $k --steps 5 --warmup 2 --min-seconds 0.5 > gpurun_out/s2/forward_${k}_kernel_stats.txt 2>&1
  head -14 gpurun_out/s2/forward_${k}_kernel_stats.txt
done
