#!/bin/bash
# samples engine clock and power while the bench runs (is the kernel power-limited?)
python bench.py --steps 400 --warmup 3 --no-cpu-baseline > /tmp/clock_probe_bench.log 2>&1 &
BP=$!
sleep 12
for i in 1 2 3 4 5; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | head -4
  sleep 0.5
done
wait $BP
tail -1 /tmp/clock_probe_bench.log | cut -c1-200
