#!/bin/bash
bash profiles/run_profiles.sh r04a_cfg5 cfg5 150 > gpurun_out/prof_r04a_cfg5.log 2>&1
python3 profiles/summarize.py gpurun_out/prof_r04a_cfg5 gpurun_out/r04a_cfg5 | grep "k_mfma_ls" | cut -c1-200
grep "k_mfma_ls" gpurun_out/r04a_cfg5_kernel_stats.csv | cut -c1-200
