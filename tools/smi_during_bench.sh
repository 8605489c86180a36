#!/bin/bash
# Sample power / clocks with rocm-smi while bench.py runs its timed loop (diagnostics: is the step power-limited?)
export TMPDIR=/tmp
python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-roofline-pass "$@" > /tmp/b.out 2> /tmp/b.err &
BP=$!
sleep 20
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks --showuse --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|GPU use|junction|hotspot" | tr '\n' ';' | sed 's/  */ /g'
  echo
  sleep 0.4
done
wait $BP
grep -E "timed" /tmp/b.err
