#!/bin/bash
# Randomised-host campaign behind profiles/r03_soak_campaign.txt: every soak under several seeds on one box; last line of each run.
seeds=${1:-"61 62 63 64 65 66"}
for seed in $seeds; do
  export SOAK_SEED=$seed
  for form in "" wave split quad mfma q4; do AIDAX_KERNEL=$form python tests/soak.py 600 2>&1 | tail -1; done
  SOAK_STREAMS=4200 python tests/soak.py 400 2>&1 | tail -1
  SOAK_STREAMS=4090 python tests/soak.py 300 2>&1 | tail -1          # one full round of 16-stream workgroups: the one-launch matrix-core forms
  SOAK_STREAMS=2600 AIDAX_KERNEL=mfma python tests/soak.py 300 2>&1 | tail -1
  SOAK_MAXF=2048 AIDAX_KERNEL=mfma python tests/soak.py 300 2>&1 | tail -1
  SOAK_STREAMS=1024 SOAK_MAXF=256 python tests/soak.py 400 2>&1 | tail -1
  python tests/soak_hub.py 1200 2>&1 | tail -1
  python tests/soak_lv2.py 1200 2>&1 | tail -1
  python tests/soak_lv2_hub.py 500 2>&1 | tail -1
done
python tests/soak_hub_rt.py 400 2>&1 | tail -1
