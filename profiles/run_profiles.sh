#!/bin/bash
# Collects the rocprofv3 evidence behind DESIGN.md on an MI355X box (run from the repo root through gpurun):
#   profiles/run_profiles.sh <tag> <workload> <steps>
# Writes gpurun_out/prof_<tag>/{trace,pmc_*}; condense with profiles/summarize.py afterwards.
# Kernel trace and each counter group are separate runs (TCC FETCH_SIZE and WRITE_SIZE do not fit one pass,
# and --pmc must not be combined with other trace domains on this pool).
set -u
tag=${1:-r01}; wl=${2:-cfg2}; steps=${3:-2000}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
py=$(command -v python3)
bench="$root/bench.py --workload $wl --steps $steps --warmup 50 --no-cpu-baseline --no-check --no-others --no-traffic --no-dist"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o r -- $py $bench > "$out/trace.log" 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$out/pmc_fetch" -o r -- $py $bench > "$out/pmc_fetch.log" 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$out/pmc_write" -o r -- $py $bench > "$out/pmc_write.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU \
    -d "$out/pmc_sq" -o r -- $py $bench > "$out/pmc_sq.log" 2>&1
if [ "$wl" = "cfg5" ] || [ "$wl" = "cfg4" ] || [ "$wl" = "cfg3" ]; then
    rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
        -d "$out/pmc_mfma" -o r -- $py $bench > "$out/pmc_mfma.log" 2>&1
fi
cd "$root"
find "$out" -name "*.csv" | head -20
for f in "$out"/*.log; do tail -n 2 "$f"; done
