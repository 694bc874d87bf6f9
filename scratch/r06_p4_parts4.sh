#!/bin/bash
cd "$(dirname "$0")/.."
run() { AIDAX_LIB=$PWD/$2 AIDAX_TUNE=$3 python bench.py --workload cfg2 --no-others --no-cpu-baseline --no-traffic --no-dist --no-check --steps 3000 --warmup 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config']['kernel'], round(d['roofline']['kernel_ms']*1e3,2), 'us')"; }
for r in 1 2; do
run "head helper idle  " scratch/prev_lib/libhead_hooks.so 786432
run "cur  helper idle  " aidadsp-lv2_amd/lib/hooks/libaidax_hip.so 786432
run "nonchk helper idle" scratch/prev_lib/libnonchk.so 786432
run "nonchk full       " scratch/prev_lib/libnonchk.so 0
done
