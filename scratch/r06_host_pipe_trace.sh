#!/bin/bash
# where a host-buffer block's time above the pass goes: kernel and copy intervals of the two-in-flight loop
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
python scratch/r06_host_pipe_trace.py 300 1
python scratch/r06_host_pipe_trace.py 300 2
python scratch/r06_host_pipe_trace.py 300 2
rm -rf gpurun_out/hp_trace; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/hp_trace -- python3 scratch/r06_host_pipe_trace.py 300 2 2>&1 | tail -1
python - <<'PY'
import csv, glob, statistics as st
k = glob.glob("gpurun_out/hp_trace/**/*kernel_trace.csv", recursive=True)[0]
m = glob.glob("gpurun_out/hp_trace/**/*memory_copy_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(k))]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:34], "q" + r["Queue_Id"]) for r in rows)
cs = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"][12:], "s" + r["Stream_Id"]) for r in csv.DictReader(open(m)))
p4 = [x for x in ks if "pipe4" in x[2]][-250:]
dur = [e - s for s, e, *_ in p4]; gap = [p4[i + 1][0] - p4[i][1] for i in range(len(p4) - 1)]
print("pass p50 %.2f us, gap between passes p50 %.2f p90 %.2f, start to start p50 %.2f" % (st.median(dur) / 1e3, st.median(gap) / 1e3, sorted(gap)[int(.9 * len(gap))] / 1e3, st.median([p4[i + 1][0] - p4[i][0] for i in range(len(p4) - 1)]) / 1e3))
t0 = p4[100][0]
for s, e, d, q in sorted(ks + cs):
    if t0 <= s < t0 + 170000: print(f"{(s - t0) / 1e3:8.2f} -> {(e - t0) / 1e3:8.2f}  ({(e - s) / 1e3:6.2f})  {d} {q}")
PY
