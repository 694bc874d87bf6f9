#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (gpurun_out/prof_*/...) into the small summaries
committed under profiles/. Usage: summarize.py <prof_dir> <out_prefix>
Expects <prof_dir>/trace/*_kernel_stats.csv and <prof_dir>/pmc_*/*_counter_collection.csv."""
import collections
import csv
import glob
import os
import sys


def main(prof, out):
    stats = glob.glob(os.path.join(prof, "trace", "*_kernel_stats.csv"))
    if stats:
        with open(stats[0]) as f, open(out + "_kernel_stats.csv", "w") as g:
            g.write(f.read())
    lines = []
    for d in sorted(glob.glob(os.path.join(prof, "pmc_*"))):
        for fn in glob.glob(os.path.join(d, "*_counter_collection.csv")):
            agg = collections.defaultdict(list)
            meta = {}
            for r in csv.DictReader(open(fn)):
                k = (r["Kernel_Name"], r["Counter_Name"])
                agg[k].append(float(r["Counter_Value"]))
                meta[r["Kernel_Name"]] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
            for (kern, ctr), v in sorted(agg.items()):
                v2 = sorted(v)
                med = v2[len(v2) // 2]
                lines.append(f"{os.path.basename(d):10s} {ctr:22s} n={len(v):4d} median={med:.6g} mean={sum(v)/len(v):.6g} min={v2[0]:.6g} max={v2[-1]:.6g}  {kern[:60]}")
            for kern, m in meta.items():
                lines.append(f"{os.path.basename(d):10s} dispatch: grid={m[0]} wg={m[1]} lds={m[2]} vgpr={m[3]} agpr={m[4]} sgpr={m[5]} scratch={m[6]}  {kern[:60]}")
    # which kernel sources these counters describe (bench.py only quotes a committed profile of the sources it runs)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    lines.insert(0, f"kernel_src_sha16={bench.kernel_sources_sha16()}  (csrc/*.hip, csrc/*.h, Makefile at the time of the summary)")
    with open(out + "_pmc_summary.txt", "w") as g:
        g.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
