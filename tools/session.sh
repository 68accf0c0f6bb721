#!/bin/bash
# final refresh of the round's profiles (one GPU session)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles
bash tools/refresh_profiles.sh r05 > gpurun_out/profiles/refresh.log 2>&1
tail -5 gpurun_out/profiles/refresh.log | cut -c1-300
for k in cfg3 cfg4 cfg5; do
  python bench.py --workload $k --steps 20 --warmup 3 > gpurun_out/profiles/r05_forward_${k}.json 2> gpurun_out/profiles/${k}.err
  tail -c 400 gpurun_out/profiles/r05_forward_${k}.json; echo
  bash tools/prof.sh r05$k python bench.py --workload $k --steps 6 --warmup 2 > gpurun_out/profiles/r05_forward_${k}_kernel_stats.txt 2>&1
done
bash tools/train_prof.sh 45 > gpurun_out/profiles/r05_train_step_body.txt 2>&1
python examples/train_synthetic.py --steps 1 2>&1 | tail -2 > gpurun_out/profiles/r05_train_wall.txt
cat gpurun_out/profiles/r05_train_wall.txt
