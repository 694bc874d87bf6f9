#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_lv2_shell.py tests/test_gpu_parity.py tests/test_gpu_lp_tenancy.py -x -q -m gpu -k "placed_over or one_layer_one_launch or tenancy or give_up or swapped" 2>&1 | tail -5
timeout 600 python scratch/coop_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_lp_coop.txt
