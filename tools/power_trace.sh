#!/bin/bash
# Samples socket power, clocks and temperature (rocm-smi) twice a second while the timed loop of bench.py runs: the direct evidence for the
# "power-limited" reading of the MFMA kernels (profiles/r03_pmc.md).  Usage: tools/power_trace.sh <out.txt> [bench.py arguments]
out=${1:-gpurun_out/power_trace.txt}; shift
mkdir -p "$(dirname "$out")"
python bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-parity-leg --timed-only "$@" > "$out.bench" 2>&1 &
bp=$!
: > "$out"
for i in $(seq 1 60); do
  if ! kill -0 $bp 2>/dev/null; then break; fi
  echo "== t=$(date +%s.%N)" >> "$out"
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp --showuse 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)|GPU use" >> "$out"
  sleep 0.5
done
wait $bp
tail -2 "$out.bench"
