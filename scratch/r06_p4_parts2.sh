#!/bin/bash
# k_lstm_pipe4 in its free-running form: what the helper's parts cost the launch (hooks build; AIDAX_TUNE 262144 = no chain passes, 524288 = no Dense; wrong output)
cd "$(dirname "$0")/.."
export AIDAX_LIB=$PWD/aidadsp-lv2_amd/lib/hooks/libaidax_hip.so AIDAX_PIPE4=1
run() { AIDAX_TUNE=$2 python bench.py --workload cfg2 --no-others --no-cpu-baseline --no-traffic --no-dist --no-check --steps 3000 --warmup 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config']['kernel'], round(d['roofline']['kernel_ms']*1e3,2), 'us')"; }
for r in 1 2; do run "full        " 0; run "no chain    " 262144; run "no Dense    " 524288; run "helper idle " 786432; done
