#!/bin/bash
# round 5, session 4: long-form attention with two query tiles per item (A/B against one), similarity products, tests of both
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s4
timeout 900 python -m pytest tests/test_r4_gpu.py -x -q -m gpu -k "inproj_attention or similarity" > gpurun_out/s4/pytest.txt 2>&1
tail -5 gpurun_out/s4/pytest.txt
timeout 900 python -m pytest tests/test_r3_gpu.py tests/test_r2_gpu.py -x -q -m gpu > gpurun_out/s4/pytest2.txt 2>&1
tail -3 gpurun_out/s4/pytest2.txt
for i in 1 2 3; do
  for v in nq1 new; do
    for k in cfg5 cfg4; do
      echo -n "$v $k "
      CENTERCLIP_HIP_LIB=$PWD/ab/lib_$v.so python bench.py --workload $k --steps 20 --warmup 3 2>/dev/null | tail -1 |
        python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
    done
  done
done > gpurun_out/s4/ab.txt 2>&1
cat gpurun_out/s4/ab.txt
python - > gpurun_out/s4/sim.json 2> gpurun_out/s4/sim.err <<'PY'
import json, torch, bench
print(json.dumps(bench.similarity_bench(torch.device("cuda", 0))))
PY
cat gpurun_out/s4/sim.json; tail -3 gpurun_out/s4/sim.err
bash tools/prof.sh fwd4_cfg5 python bench.py --workload cfg5 --steps 5 --warmup 2 --min-seconds 0.5 > gpurun_out/s4/forward_cfg5_kernel_stats.txt 2>&1
head -8 gpurun_out/s4/forward_cfg5_kernel_stats.txt
