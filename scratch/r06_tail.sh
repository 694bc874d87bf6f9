#!/bin/bash
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r06_gpu_suite.txt
cat gpurun_out/r06_gpu_suite.txt
