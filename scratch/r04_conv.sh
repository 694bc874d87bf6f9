#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -x -q -m gpu -k "conv or cfg4 or extension_models or every_kernel_form" 2>&1 | tail -4
python scratch/conv_forms.py 2>&1 | grep -v amdgpu | head -12
bash profiles/run_profiles.sh r04c_cfg4 cfg4 1500 > gpurun_out/prof_r04c_cfg4.log 2>&1
python3 profiles/summarize.py gpurun_out/prof_r04c_cfg4 gpurun_out/r04c_cfg4 | grep -E "SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE" | cut -c1-150
head -3 gpurun_out/r04c_cfg4_kernel_stats.csv | cut -c1-150
