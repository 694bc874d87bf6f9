#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -x -q -m gpu -k "gru or cfg3 or gate_major or reference_variant or drift" 2>&1 | tail -6
run() { AIDAX_LIB=$2 python bench.py --workload cfg3 --no-others --no-cpu-baseline --no-traffic --no-dist --steps 1500 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config']['kernel'], round(d['roofline']['kernel_ms']*1e3,2), 'us', d['max_abs_err'])"; }
for round in 1 2 3; do run r5 $PWD/scratch/prev_lib/libaidax_r5_ship.so; run r6 $PWD/aidadsp-lv2_amd/lib/libaidax_hip.so; done
