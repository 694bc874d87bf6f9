#!/bin/bash
# review item 6 (cfg5): upper bounds before building anything — hooks build, AIDAX_TUNE 2097152 = k_mfma_ls reads ONE h fragment per tick and wave instead of nine (wrong output)
cd "$(dirname "$0")/.."
export AIDAX_LIB=$PWD/aidadsp-lv2_amd/lib/hooks/libaidax_hip.so
run() { AIDAX_TUNE=$2 python bench.py --workload cfg5 --no-others --no-cpu-baseline --no-traffic --no-check --steps 400 --warmup 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('gpu_state_while_running') or {}; print('$1', d['config']['kernel'], round(d['roofline']['kernel_ms']*1e3,1), 'us', g.get('sclk_mhz'), 'MHz', g.get('socket_power_w'), 'W')"; }
for round in 1 2; do run "nine fragment reads " 0; run "one fragment read   " 2097152; done
