python bench.py --no-cpu-baseline --no-extras --steps 25000 > /tmp/b.log 2>&1 &
BP=$!
sleep 40
for i in 1 2 3 4; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|edge)" | head -8; echo ---; sleep 2; done
wait $BP; tail -1 /tmp/b.log | cut -c1-200
