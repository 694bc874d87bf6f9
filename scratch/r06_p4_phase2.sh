#!/bin/bash
# k_lstm_pipe4 in its free-running form (the tree's): recurrent wave j waits j x nd x 16 cycles ONCE before its first tile (run-time parameter, one binary)
cd "$(dirname "$0")/.."
export AIDAX_LIB=$PWD/aidadsp-lv2_amd/lib/hooks/libaidax_hip.so AIDAX_PIPE4=1
run() { AIDAX_TUNE=$(( $1 << 24 )) python bench.py --workload cfg2 --no-others --no-cpu-baseline --no-traffic --no-dist --no-check --steps 3000 --warmup 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nd=$1', d['config']['kernel'], round(d['roofline']['kernel_ms']*1e3,2), 'us')"; }
for nd in 0 1 2 3 4 5 6 8 10 12 15 0; do run $nd; done
AIDAX_PIPE4=0 run 0
