#!/bin/bash
# Round 6's randomised-host campaign on the final tree (tests/soak*.py against the oracle's plugin mirror): the table models of a pool now run k_*_pipe4
# for whole-tile blocks (conditioned ones at every pool size) and k_*_pipe for the ragged ones in between — one state; a one-stream pool's pass writes
# its own completion word. Last line of each run. Shipped library unless a form is forced (then the hooks build).
cd "$(dirname "$0")/.."
HOOKS=aidadsp-lv2_amd/lib/hooks/libaidax_hip.so
seeds=${1:-"91 92 93"}
for seed in $seeds; do
  export SOAK_SEED=$seed
  python tests/soak.py 600 2>&1 | tail -1
  SOAK_STREAMS=1 python tests/soak.py 600 2>&1 | tail -1
  SOAK_STREAMS=3 python tests/soak.py 400 2>&1 | tail -1
  AIDAX_KERNEL_WORD=0 SOAK_STREAMS=1 python tests/soak.py 300 2>&1 | tail -1
  SOAK_STREAMS=1024 SOAK_MAXF=256 python tests/soak.py 400 2>&1 | tail -1
  SOAK_STREAMS=4200 python tests/soak.py 300 2>&1 | tail -1
  SOAK_MAXF=2048 python tests/soak.py 300 2>&1 | tail -1
  SOAK_SR=44100 SOAK_STREAMS=300 python tests/soak.py 300 2>&1 | tail -1
  AIDAX_LIB=$HOOKS AIDAX_PIPE4=0 python tests/soak.py 300 2>&1 | tail -1
  AIDAX_LIB=$HOOKS AIDAX_CONV_ST=0 python tests/soak.py 300 2>&1 | tail -1
  python tests/soak_hub.py 800 2>&1 | tail -1
  python tests/soak_lv2.py 800 2>&1 | tail -1
  python tests/soak_lv2_hub.py 300 2>&1 | tail -1
done
python tests/soak_hub_rt.py 300 2>&1 | tail -1
python tests/fuzz_abi.py 40 2>&1 | tail -1
