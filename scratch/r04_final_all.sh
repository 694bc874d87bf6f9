#!/bin/bash
mkdir -p gpurun_out
timeout 2900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r04_full_gpu_tests.txt
./scratch/r04_final_prof.sh r04f
