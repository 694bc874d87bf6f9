#!/bin/bash
# k_conv_st (the streaming form of cfg4's full blocks): the conv parity tests on the hooks build, then cfg4 through bench.py with and without it
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conv" 2>&1 | tail -3
export AIDAX_LIB=$PWD/aidadsp-lv2_amd/lib/hooks/libaidax_hip.so
run() { python bench.py --workload cfg4 --no-others --no-cpu-baseline --no-traffic --steps 2000 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config']['kernel'], round(d['ms_per_step']*1e3,2), 'us', d['max_abs_err'], {k: round(v,1) for k,v in d['per_launch_us'].items()}, d['gpu_state_while_running']['socket_power_w'], 'W', d['gpu_state_while_running']['sclk_mhz'], 'MHz')"; }
for v in "$@"; do env $v bash -c "$(declare -f run); run '$v'"; done
