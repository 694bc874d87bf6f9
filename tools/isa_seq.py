#!/usr/bin/env python3
"""Instruction-class sequence of a gfx950 kernel, basic block by basic block.

    python3 tools/isa_seq.py <file.hip> <kernel-name-regex> [--min N] [-- extra hipcc flags]

Compiles the source to device assembly with the product's flags (Makefile: HIPFLAGS) and prints, for every basic
block of the matching kernels with at least N instructions, a run-length string over instruction classes:

    M  v_mfma_*            v  other VALU          t  transcendental VALU (v_exp/v_rcp/v_rsq/v_log/v_sqrt/v_sin/v_cos)
    d  ds_read / ds_load   D  ds_write / ds_store g  global/buffer/flat load    G  global/buffer/flat store
    w  s_waitcnt           n  s_nop               b  s_barrier                  s  other SALU   p  s_setprio   B  branch

e.g. "M12 d4 w M12" = twelve MFMAs, four LDS reads, a wait, twelve MFMAs. Used to check that what a tick loop issues
is what the source meant (sched_group_barrier interleaves, compiler-added s_nop / s_waitcnt).
"""
import re
import subprocess
import sys

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-Iinclude",
         "-Iaidadsp-lv2_amd/csrc", "--cuda-device-only", "-S", "-o", "-"]
TRANS = re.compile(r"^v_(exp|rcp|rsq|log|sqrt|sin|cos)_")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "M"
    if op.startswith("v_"):
        return "t" if TRANS.match(op) else "v"
    if op.startswith("ds_"):
        return "D" if ("write" in op or "store" in op) else "d"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "G" if "store" in op else "g"
    if op == "s_waitcnt":
        return "w"
    if op == "s_nop":
        return "n"
    if op == "s_barrier":
        return "b"
    if op == "s_setprio":
        return "p"
    if op.startswith(("s_cbranch", "s_branch")):
        return "B"
    if op.startswith("s_"):
        return "s"
    return "?"


def rle(seq):
    out, i = [], 0
    while i < len(seq):
        j = i
        while j < len(seq) and seq[j] == seq[i]:
            j += 1
        out.append(seq[i] + (str(j - i) if j - i > 1 else ""))
        i = j
    return " ".join(out)


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        k = args.index("--")
        args, extra = args[:k], args[k + 1:]
    min_n = 24
    if "--min" in args:
        k = args.index("--min")
        min_n = int(args[k + 1])
        del args[k:k + 2]
    src, pat = args[0], re.compile(args[1])
    asm = subprocess.run(["hipcc"] + FLAGS + extra + [src], check=True, capture_output=True, text=True).stdout
    kernel, block, seq = None, None, []

    def flush():
        if kernel and pat.search(kernel) and len(seq) >= min_n:
            counts = {c: seq.count(c) for c in "Mvtdwnb"}
            print(f"  {block:<14} {len(seq):5d} instr  " + " ".join(f"{c}={n}" for c, n in counts.items() if n))
            print("      " + rle(seq))

    for line in asm.splitlines():
        s = line.strip()
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", s)
        if m:
            flush()
            seq = []
            name = m.group(1)
            if not name.startswith(".L"):
                kernel = name
                if pat.search(kernel):
                    dem = subprocess.run(["c++filt", kernel], capture_output=True, text=True).stdout.strip()
                    print(f"{dem}")
            block = name
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        if s.startswith("s_endpgm"):
            flush()
            seq = []
            continue
        seq.append(classify(s.split()[0]))
    flush()


if __name__ == "__main__":
    main()
