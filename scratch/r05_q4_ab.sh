#!/bin/bash
# round-4 review item 5, the structural A/B: cfg2 through k_lstm_q4 (four streams of a CU as the columns of v_mfma_f32_4x4x1) against k_lstm_pipe<32>,
# on the hooks build (AIDAX_KERNEL forces the form); parts of q4 switched off by AIDAX_TUNE (32 = helper waves idle, 64 = cell waves idle, 128 = no Q, 256 = no P)
export AIDAX_LIB=$PWD/aidadsp-lv2_amd/lib/hooks/libaidax_hip.so
run() { python bench.py --steps 3000 --warmup 50 --no-cpu-baseline --no-others --no-traffic --no-dist 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['config']['kernel'], round(d['roofline']['kernel_ms']*1e3,2), 'us', 'err', d.get('max_abs_err'))"; }
AIDAX_KERNEL=pipe run pipe
AIDAX_KERNEL=q4 run q4
AIDAX_KERNEL=q4 AIDAX_TUNE=32 run q4_no_helpers
AIDAX_KERNEL=q4 AIDAX_TUNE=64 run q4_no_cell
AIDAX_KERNEL=q4 AIDAX_TUNE=96 run q4_barriers_only
AIDAX_KERNEL=q4 AIDAX_TUNE=128 run q4_no_Q
AIDAX_KERNEL=q4 AIDAX_TUNE=256 run q4_no_P
AIDAX_KERNEL=pipe run pipe_again
