#!/bin/bash
mkdir -p gpurun_out
timeout 1700 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "layer_pipelined or stacked_one_launch or (full_size and cfg5) or random_recurrent" 2>&1 | tail -15 | tee gpurun_out/r04_ls_tests.txt
python scratch/ls_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_ls_ab.txt
