python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r3_t7_tests.txt
python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r3_t7_bench.json 2> gpurun_out/r3_t7_bench.err
cat gpurun_out/r3_t7_tests.txt; tail -c 400 gpurun_out/r3_t7_bench.json
