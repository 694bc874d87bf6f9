#!/bin/bash
timeout 2900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python scratch/gs_threshold.py 2>&1 | grep -v amdgpu.ids | awk -F'|' '{print $1 "|" $2}' | tee gpurun_out/r04_gs_threshold_after.txt
