#!/bin/bash
# Round 5's randomised-host campaign (tests/soak*.py: random block sizes, control flips, activates, model swaps across kernel families — the conv
# stack now on k_conv_st for full blocks and k_conv_ms for ragged ones, one state — against the oracle's plugin mirror); last line of each run. Shipped library unless a form is forced (then the hooks build).
cd "$(dirname "$0")/.."
HOOKS=aidadsp-lv2_amd/lib/hooks/libaidax_hip.so
seeds=${1:-"71 72 73 74"}
for seed in $seeds; do
  export SOAK_SEED=$seed
  python tests/soak.py 600 2>&1 | tail -1
  SOAK_STREAMS=1024 SOAK_MAXF=256 python tests/soak.py 400 2>&1 | tail -1
  SOAK_STREAMS=4200 python tests/soak.py 300 2>&1 | tail -1
  SOAK_MAXF=2048 python tests/soak.py 300 2>&1 | tail -1
  SOAK_SR=44100 SOAK_STREAMS=300 python tests/soak.py 300 2>&1 | tail -1
  AIDAX_LIB=$HOOKS AIDAX_CONV_MS=0 python tests/soak.py 300 2>&1 | tail -1
  AIDAX_LIB=$HOOKS AIDAX_CONV_FUSED=0 python tests/soak.py 300 2>&1 | tail -1
  AIDAX_LIB=$HOOKS AIDAX_CONV_ST=0 python tests/soak.py 300 2>&1 | tail -1
  python tests/soak_hub.py 800 2>&1 | tail -1
  python tests/soak_lv2.py 800 2>&1 | tail -1
  python tests/soak_lv2_hub.py 300 2>&1 | tail -1
done
