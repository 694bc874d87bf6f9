#!/bin/bash
# round 5, final evidence after k_conv_st: the GPU suite, smoke, the profiles of every BASELINE config, the bench line as the driver runs it, cfg4's timeline
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r05_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r05_gpu_suite.txt 2>&1
cat gpurun_out/r05_gpu_suite.txt
bash scratch/r05_final_prof.sh r05 2>&1 | tail -30
AIDAX_LIB=$PWD/build/lib_cv/libaidax_hip.so python scratch/st_trace.py > gpurun_out/r05_st_trace.txt 2>&1
tail -8 gpurun_out/r05_st_trace.txt
AIDAX_LIB=$PWD/build/lib_pt/libaidax_hip.so python scratch/pipe_frame.py > gpurun_out/r05_pipe_frame.txt 2>&1
tail -30 gpurun_out/r05_pipe_frame.txt
