#!/bin/bash
# k_lstm_pipe<32>'s recurrent stage as a loop of R frames per iteration (measurement builds -DAIDAX_PIPE_ROLL=R, hooks) against the sixteen unrolled
# frames of the tree; each also with the helper waves idle (AIDAX_TUNE 786432: wrong output)
cd "$(dirname "$0")/.."
run() { AIDAX_LIB=$PWD/$2 AIDAX_TUNE=$3 python bench.py --workload cfg2 --no-others --no-cpu-baseline --no-traffic --no-dist --no-check --steps 3000 --warmup 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config']['kernel'], round(d['roofline']['kernel_ms']*1e3,2), 'us')"; }
for rep in 1 2; do
run "unrolled 16        " aidadsp-lv2_amd/lib/hooks/libaidax_hip.so 0
run "unrolled 16, N only" aidadsp-lv2_amd/lib/hooks/libaidax_hip.so 786432
for r in 8 4 2 1; do run "loop of $r          " scratch/prev_lib/libroll$r.so 0; run "loop of $r, N only  " scratch/prev_lib/libroll$r.so 786432; done
done
