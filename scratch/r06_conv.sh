#!/bin/bash
mkdir -p gpurun_out/r06_conv
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -x -q -m gpu -k "conv or extension" 2>&1 | tail -25 > gpurun_out/r06_conv/tests.txt
cat gpurun_out/r06_conv/tests.txt
timeout 600 python scratch/r06_conv_forms.py > gpurun_out/r06_conv/forms.txt 2>&1
cat gpurun_out/r06_conv/forms.txt
