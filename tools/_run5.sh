python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3_t5_tests.txt
python tools/sel_prof.py > gpurun_out/r3_t5_selprof.txt 2>&1
python tools/time_cluster.py > gpurun_out/r3_t5_cluster.txt 2>&1
python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r3_t5_bench.json 2> gpurun_out/r3_t5_bench.err
bash tools/insitu.sh r3t5 > /dev/null 2>&1
cat gpurun_out/r3_t5_tests.txt gpurun_out/r3_t5_selprof.txt gpurun_out/r3_t5_cluster.txt; grep -i "select\|gram\|reduce_tok" gpurun_out/insitu_r3t5.txt
