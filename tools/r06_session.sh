#!/bin/bash
# Round-6 measurement session (one gpurun call): GPU suite, smoke, bench --pmc (counters), the default bench line, rocprofv3
# kernel stats of the same command for cfg 2 and of the step alone for cfg 3 / 4 / 5.  Results -> gpurun_out/r06/.
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $o/gputest.log; cat $o/gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --pmc --steps 20 > $o/bench_pmc_line.json 2> $o/bench_pmc.err
cp gpurun_out/bench_pmc_counters.json $o/bench_pmc_counters.json 2>/dev/null
python bench.py > $o/bench_n1_line.json 2> $o/bench_n1.err
cp gpurun_out/bench_detail_n1.json $o/bench_detail_n1.json
tail -c 2600 $o/bench_n1_line.json; echo
bash tools/prof.sh r06 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $o/r06_bench_kernel_stats.txt 2>&1
cp $(find gpurun_out/prof_r06 -name "*kernel_stats.csv" | head -1) $o/r06_bench_kernel_stats.csv
head -14 $o/r06_bench_kernel_stats.txt | cut -c1-130
for w in cfg3 cfg4 cfg5; do
  bash tools/prof.sh r06_$w python bench.py --workload $w --steps 10 --warmup 2 > $o/r06_forward_${w}_kernel_stats.txt 2>&1
  head -8 $o/r06_forward_${w}_kernel_stats.txt | cut -c1-130
done
