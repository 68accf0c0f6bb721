python -m pytest tests -m gpu -x -q -k "cluster or kmedoids or p0 or p1 or p2 or dup or perm or c1 or c4 or variant or spectral or timed or paired or smoke or s1 or grad" 2>&1 | tail -8 > gpurun_out/r3_t6_tests.txt
python tools/sel_prof.py > gpurun_out/r3_t6_selprof.txt 2>&1
python tools/time_cluster.py > gpurun_out/r3_t6_cluster.txt 2>&1
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r3_t6_bench.json 2> gpurun_out/r3_t6_bench.err
bash tools/insitu.sh r3t6 > /dev/null 2>&1
cat gpurun_out/r3_t6_tests.txt gpurun_out/r3_t6_selprof.txt gpurun_out/r3_t6_cluster.txt; grep -i "select\|gram\|reduce_tok" gpurun_out/insitu_r3t6.txt; tail -c 300 gpurun_out/r3_t6_bench.json
