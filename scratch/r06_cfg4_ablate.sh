#!/bin/bash
# Round 6, review item 3: what are cfg4's 40 us made of? One switch at a time on the hooks build (AIDAX_TUNE bits of k_conv_st):
#   32768  no history stores          65536  every far tap reads one resident tile (L2) instead of its trip to HBM
# time (HIP events, 3000 launches), clock / power while running (rocm-smi), FETCH_SIZE / WRITE_SIZE per launch (rocprofv3 --pmc, own passes)
root=$(pwd); out=$root/gpurun_out/r06_ablate; mkdir -p $out
export AIDAX_LIB=$root/aidadsp-lv2_amd/lib/hooks/libaidax_hip.so
export TMPDIR=/tmp
py=$(command -v python3)
for spec in "0 1024" "32768 1024" "65536 1024" "98304 1024" "0 512" "0 768"; do
  set -- $spec; tune=$1; streams=$2
  export AIDAX_TUNE=$tune
  tag=t${tune}_s${streams}
  $py bench.py --workload cfg4 --streams $streams --steps 3000 --warmup 100 --no-others --no-cpu-baseline --no-traffic --no-check > $out/$tag.json 2> $out/$tag.err
  cd /tmp
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --output-format csv --pmc $ctr -d $out/pmc_${tag}_$ctr -o r -- $py $root/bench.py --workload cfg4 --streams $streams --steps 300 --warmup 30 --no-others --no-cpu-baseline --no-traffic --no-check --no-dist > $out/pmc_${tag}_$ctr.log 2>&1
  done
  cd $root
  $py - $out $tag $tune $streams <<'PY'
import sys, json, glob, csv
out, tag, tune, streams = sys.argv[1:5]
d = json.loads(open(f"{out}/{tag}.json").read().strip().splitlines()[-1])
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    got = []
    for fn in glob.glob(f"{out}/pmc_{tag}_{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == ctr and "k_conv_st" in r["Kernel_Name"]:
                got.append(float(r["Counter_Value"]))
    got.sort()
    vals[ctr] = got[len(got) // 2] if got else float("nan")
g = d.get("gpu_state_while_running") or {}
pl = d.get("per_launch_us") or {}
print(f"tune={tune:>6s} streams={streams:>5s}  kernel {d['roofline']['kernel_ms']*1e3:6.2f} us  p50 {pl.get('p50', float('nan')):6.2f}  sclk {g.get('sclk_mhz')} MHz  {g.get('socket_power_w')} W   FETCH {vals['FETCH_SIZE']/1024:7.2f} MB (x2 = {2*vals['FETCH_SIZE']/1024:7.2f})  WRITE {vals['WRITE_SIZE']/1024:7.2f} MB   {d['config']['kernel']}")
PY
done
