#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gate_major or long_run_drift_gru64 or (full_size and cfg3)" 2>&1 | tail -15 > gpurun_out/r04_gs_tests.txt
cat gpurun_out/r04_gs_tests.txt
python scratch/gs_ab.py 2>&1 | tee gpurun_out/r04_gs_ab.txt
cat gpurun_out/parity_errors.json | tr ',' '\n' | grep -i "gru\|drift"
