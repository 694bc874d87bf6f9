#!/bin/bash
# randomised hosts on the kernels that came late in round 4 (k_lstm_gs, k_mfma_ls1, k_mfma_ls in ranges): tests/soak.py with SOAK_MODELS=wide
for seed in ${1:-"71 72 73 74"}; do
  export SOAK_SEED=$seed SOAK_MODELS=wide
  SOAK_STREAMS=4200 python tests/soak.py 300 2>&1 | tail -1
  SOAK_STREAMS=4090 python tests/soak.py 300 2>&1 | tail -1
  SOAK_STREAMS=2600 python tests/soak.py 300 2>&1 | tail -1
  SOAK_STREAMS=1500 SOAK_MAXF=1024 python tests/soak.py 200 2>&1 | tail -1
  SOAK_STREAMS=8200 python tests/soak.py 150 2>&1 | tail -1
done
