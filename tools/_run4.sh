python -m pytest tests -m gpu -x -q -k "similar or eval or s3 or S3 or nccl or packed or bench_line or sim or linear or gemm" 2>&1 | tail -8 > gpurun_out/r3_t4_tests.txt
python tools/sim_tile.py 2>&1 | tail -1 > gpurun_out/r3_t4_simtile.txt
python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r3_t4_bench.json 2> gpurun_out/r3_t4_bench.err
bash tools/prof.sh r3t4 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3_t4_stats.txt 2>&1
cat gpurun_out/r3_t4_tests.txt gpurun_out/r3_t4_simtile.txt; head -8 gpurun_out/r3_t4_stats.txt
