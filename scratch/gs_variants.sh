#!/bin/bash
# k_gru_gs: parity, tick trace of the current build, then A/B of the measurement builds (scratch/prev_lib/libaidax_<v>.so)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gate_major or (full_size and cfg3)" 2>&1 | tail -3
AIDAX_LIB=scratch/prev_lib/libaidax_tr.so python scratch/gs_trace.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_gs_trace.txt
for v in "" $GS_VARIANTS; do
  echo "== build: ${v:-head}"
  if [ -n "$v" ]; then export AIDAX_LIB=scratch/prev_lib/libaidax_$v.so; else unset AIDAX_LIB; fi
  GS_ONLY=1 python scratch/gs_ab.py 2>&1 | grep GRU
done 2>&1 | tee gpurun_out/r04_gs_variants.txt
