#!/bin/bash
# k_lstm_pipe4 in its barrier form (measurement build -DAIDAX_P4_BARRIER, test hooks): ONE binary, the phase between the waves as a run-time
# parameter — hd = sixteen-cycle steps the helper waits behind every barrier, nd = steps recurrent wave j waits, times j
cd "$(dirname "$0")/.."
export AIDAX_LIB=$PWD/scratch/prev_lib/libp4_bar.so AIDAX_PIPE4=1
run() { AIDAX_TUNE=$(( ($1 << 20) | ($2 << 24) )) python bench.py --workload cfg2 --no-others --no-cpu-baseline --no-traffic --no-dist --no-check --steps 3000 --warmup 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hd=$1 nd=$2', d['config']['kernel'], round(d['roofline']['kernel_ms']*1e3,2), 'us')"; }
for hd in 0 1 2 3 4 6 8 10 12 15; do run $hd 0; done
for nd in 1 2 3; do run 0 $nd; run 4 $nd; done
run 0 0
