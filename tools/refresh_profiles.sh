#!/bin/bash
# One GPU session that regenerates everything under profiles/ for round $1 (default r01):
#   PMC HBM-traffic and MFMA-utilisation passes (each counter group in its own run, counters + kernel trace only),
#   then the bench line (with extras + cpu baseline; it reads the fresh traffic file) and the rocprofv3 kernel
#   stats of the same bench command.  Results land in gpurun_out/profiles/ - copy them into profiles/.
r=${1:-r04}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles
bash tools/pmc.sh $r python tools/kernels_for_pmc.py > gpurun_out/profiles/pmc.log 2>&1
cp gpurun_out/pmc_$r/traffic.json gpurun_out/profiles/${r}_traffic_pmc.json
cp gpurun_out/pmc_$r/traffic.json profiles/${r}_traffic_pmc.json
tail -8 gpurun_out/profiles/pmc.log
bash tools/pmc_mfma.sh > gpurun_out/profiles/mfma.log 2>&1
cp gpurun_out/pmc_mfma/mfma_util.json gpurun_out/profiles/${r}_mfma_util_pmc.json
tail -6 gpurun_out/profiles/mfma.log | cut -c1-220
python bench.py --steps 30 --warmup 3 > gpurun_out/profiles/${r}_bench_n1.json 2> gpurun_out/profiles/bench.err
tail -c 600 gpurun_out/profiles/${r}_bench_n1.json
bash tools/prof.sh $r python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/profiles/${r}_bench_kernel_stats.txt 2>&1
cp $(find gpurun_out/prof_$r -name "*kernel_stats.csv" | head -1) gpurun_out/profiles/${r}_bench_kernel_stats.csv
head -12 gpurun_out/profiles/${r}_bench_kernel_stats.txt
