#!/bin/bash
# round 4, first GPU call: the corrected overlap measurement, the RCCL branch at world size 1, the long-run drift pins
mkdir -p gpurun_out
./scratch/uoverlap2 1 > gpurun_out/r04_overlap_1wg.txt 2>&1
./scratch/uoverlap2 256 > gpurun_out/r04_overlap_256wg.txt 2>&1
timeout 1500 python -m pytest tests/test_dist_gloo.py tests/test_gpu_parity.py tests/test_lv2_shell.py -x -q -m gpu -k "rccl or long_run or equal_slices" 2>&1 | tail -15 > gpurun_out/r04_first_tests.txt
cat gpurun_out/r04_overlap_1wg.txt
tail -12 gpurun_out/r04_overlap_256wg.txt
cat gpurun_out/r04_first_tests.txt
