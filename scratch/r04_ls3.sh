#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "layer_pipelined or stacked_one_launch or (full_size and cfg5) or random_recurrent" 2>&1 | tail -4
python scratch/ls_ab.py 2>&1 | grep "x2\|x3" | tee gpurun_out/r04_ls_ab.txt
