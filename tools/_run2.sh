python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3_t2_tests.txt
python -m pytest tests/test_r3_gpu.py -m gpu -q -s 2>&1 | grep "^\[" > gpurun_out/r3_t2_r3detail.txt
python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r3_t2_bench.json 2> gpurun_out/r3_t2_bench.err
bash tools/prof.sh r3t2 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3_t2_stats.txt 2>&1
cat gpurun_out/r3_t2_tests.txt gpurun_out/r3_t2_r3detail.txt; head -14 gpurun_out/r3_t2_stats.txt
