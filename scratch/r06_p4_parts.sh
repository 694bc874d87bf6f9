#!/bin/bash
# what the helper wave of k_lstm_pipe4 costs the recurrent waves: hooks build, AIDAX_TUNE 262144 = no chain passes, 524288 = no Dense (wrong output), AIDAX_PIPE4=0 = k_lstm_pipe
cd "$(dirname "$0")/.."
export AIDAX_LIB=$PWD/aidadsp-lv2_amd/lib/hooks/libaidax_hip.so
run() { AIDAX_PIPE4=$2 AIDAX_TUNE=$3 python bench.py --workload cfg2 --no-others --no-cpu-baseline --no-traffic --no-dist --no-check --steps 4000 --warmup 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config']['kernel'], round(d['roofline']['kernel_ms']*1e3,2), 'us')"; }
for round in 1 2; do
run "pipe  full            " 0 0
run "pipe  P,Q idle        " 0 786432
run "pipe4 full            " 1 0
run "pipe4 no chain        " 1 262144
run "pipe4 no Dense        " 1 524288
run "pipe4 helper idle     " 1 786432
done
