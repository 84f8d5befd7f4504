#!/bin/bash
# round 6, first product call: the msda stress test on the fixed library, the branch overlap A/B, parity of the overlapped path
mkdir -p gpurun_out
{
echo "== stress test"; timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "msda" 2>&1 | tail -3
echo "== bench, one stream"; HIPIE_BRANCH_STREAMS=0 timeout 600 python bench.py --steps 10 --warmup 3 --timed-only 2>&1 | tail -1
echo "== bench, two streams"; timeout 600 python bench.py --steps 10 --warmup 3 --timed-only 2>&1 | tail -1
echo "== bench, one stream again"; HIPIE_BRANCH_STREAMS=0 timeout 600 python bench.py --steps 10 --warmup 3 --timed-only 2>&1 | tail -1
echo "== bench, two streams again"; timeout 600 python bench.py --steps 10 --warmup 3 --timed-only 2>&1 | tail -1
echo "== e2e parity, two streams"; timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x 2>&1 | tail -3
} > gpurun_out/r6_call1.log 2>&1
