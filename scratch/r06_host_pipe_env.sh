#!/bin/bash
# the staged host pipeline (upload | pass | download on three queues) under the runtime's copy-path switches
cd "$(dirname "$0")/.."; export TMPDIR=/tmp AIDAX_HOST_STREAM=0
echo "default:"; python scratch/r06_host_pipe_trace.py 300 2
echo "GPU_FORCE_BLIT_COPY_SIZE=0:"; GPU_FORCE_BLIT_COPY_SIZE=0 python scratch/r06_host_pipe_trace.py 300 2
echo "DEBUG_CLR_LIMIT_BLIT_WG=8:"; DEBUG_CLR_LIMIT_BLIT_WG=8 python scratch/r06_host_pipe_trace.py 300 2
echo "DEBUG_CLR_LIMIT_BLIT_WG=32:"; DEBUG_CLR_LIMIT_BLIT_WG=32 python scratch/r06_host_pipe_trace.py 300 2
echo "GPU_BLIT_ENGINE_TYPE=2:"; GPU_BLIT_ENGINE_TYPE=2 python scratch/r06_host_pipe_trace.py 300 2
echo "GPU_STREAMOPS_CP_WAIT=1:"; GPU_STREAMOPS_CP_WAIT=1 python scratch/r06_host_pipe_trace.py 300 2
echo "AMD_DIRECT_DISPATCH=0:"; AMD_DIRECT_DISPATCH=0 python scratch/r06_host_pipe_trace.py 300 2
echo "HSA_ENABLE_SDMA=0:"; HSA_ENABLE_SDMA=0 python scratch/r06_host_pipe_trace.py 300 2
