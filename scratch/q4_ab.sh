#!/bin/bash
# cfg2 through k_lstm_q4 with parts switched off (AIDAX_TUNE: 32 = helper waves idle, 64 = cell waves idle) next to k_lstm_pipe
run() { python bench.py --steps 3000 --warmup 50 --no-cpu-baseline --no-others --no-traffic --no-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['config']['kernel'], round(d['roofline']['kernel_ms']*1e3,2), 'us')"; }
AIDAX_KERNEL=pipe run pipe
run q4
AIDAX_TUNE=32 run q4_no_helpers
AIDAX_TUNE=64 run q4_no_cell
AIDAX_TUNE=96 run q4_barriers_only
AIDAX_TUNE=128 run q4_no_Q
AIDAX_TUNE=256 run q4_no_P
