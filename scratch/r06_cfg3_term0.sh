#!/bin/bash
# review item 8 (cfg3: publish term 0 of h first): an upper bound before building it — hooks build, AIDAX_TUNE 1048576 = k_gru_gs publishes term 0 only (wrong output)
cd "$(dirname "$0")/.."
export AIDAX_LIB=$PWD/aidadsp-lv2_amd/lib/hooks/libaidax_hip.so
run() { AIDAX_TUNE=$2 python bench.py --workload cfg3 --no-others --no-cpu-baseline --no-traffic --no-dist --no-check --steps 1500 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config']['kernel'], round(d['roofline']['kernel_ms']*1e3,2), 'us')"; }
for round in 1 2 3; do run "full split      " 0; run "term 0 only     " 1048576; done
