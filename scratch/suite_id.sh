#!/bin/bash
# the GPU suite with the box's identity and the complete failure log (looking for a box-dependent failure)
tag=$1
{ hostname; rocm-smi --showuniqueid --showclocks --showperflevel 2>&1 | grep -iE "unique|sclk|mclk|perf" | head -8; } > gpurun_out/suite_${tag}_box.txt
timeout 900 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -300 > gpurun_out/suite_${tag}.txt
tail -2 gpurun_out/suite_${tag}.txt; cat gpurun_out/suite_${tag}_box.txt
