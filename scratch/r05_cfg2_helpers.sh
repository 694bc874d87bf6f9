#!/bin/bash
# what the helper waves of k_lstm_pipe<32> cost the recurrent wave: cfg2 with wave P idle (AIDAX_TUNE 262144), wave Q idle (524288), both (wrong output by design)
cd "$(dirname "$0")/.."
export AIDAX_LIB=$PWD/aidadsp-lv2_amd/lib/hooks/libaidax_hip.so
run() { AIDAX_TUNE=$2 python bench.py --workload cfg2 --no-others --no-cpu-baseline --no-traffic --no-dist --no-check --steps 3000 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config']['kernel'], round(d['roofline']['kernel_ms']*1e3,2), 'us')"; }
for r in 1 2; do run all 0; run P_idle 262144; run Q_idle 524288; run both_idle 786432; done
