#!/bin/bash
export HIPIE_MIOPEN_FIND=0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_e2e.py -q -m gpu -x -k "r50 or full_size" 2>&1 | tail -6
timeout 300 python tools/stage_times.py split3 shapes > gpurun_out/stage_c.txt 2>&1; head -70 gpurun_out/stage_c.txt
timeout 300 bash tools/power_trace.sh gpurun_out/power_trace.txt; grep -c "==" gpurun_out/power_trace.txt; sed -n 1,12p gpurun_out/power_trace.txt; tail -12 gpurun_out/power_trace.txt
timeout 400 python bench.py --steps 3 --warmup 1 --classes 150 --text-len 815 --no-cpu-baseline --no-parity-leg > gpurun_out/cfg3.log 2>&1; tail -c 700 gpurun_out/cfg3.log
timeout 400 python bench.py --steps 3 --warmup 1 --classes 1203 --text-len 4096 --size 1344 --no-cpu-baseline --no-parity-leg > gpurun_out/cfg4.log 2>&1; tail -c 700 gpurun_out/cfg4.log
