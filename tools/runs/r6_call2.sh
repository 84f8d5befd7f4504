#!/bin/bash
# round 6: branch overlap A/B on one box (alternating), JSON lines only
mkdir -p gpurun_out
{
for i in 1 2; do
echo "== one stream"; HIPIE_BRANCH_STREAMS=0 timeout 600 python bench.py --steps 10 --warmup 3 --timed-only 2>/dev/null | tail -1
echo "== two streams"; timeout 600 python bench.py --steps 10 --warmup 3 --timed-only 2>/dev/null | tail -1
done
} > gpurun_out/r6_call2.log 2>&1
