python -m pytest tests -m gpu -x -q -k "similar or eval or s3 or S3 or nccl or packed or bench_line or sim" 2>&1 | tail -8 > gpurun_out/r3_t3_tests.txt
python tools/sim_tile.py > gpurun_out/r3_t3_simtile.txt 2>&1
python tools/sel_prof.py > gpurun_out/r3_t3_selprof.txt 2>&1
python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r3_t3_bench.json 2> gpurun_out/r3_t3_bench.err
bash tools/prof.sh r3t3 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3_t3_stats.txt 2>&1
cat gpurun_out/r3_t3_tests.txt gpurun_out/r3_t3_simtile.txt gpurun_out/r3_t3_selprof.txt; head -8 gpurun_out/r3_t3_stats.txt
