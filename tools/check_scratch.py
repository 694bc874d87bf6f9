#!/usr/bin/env python3
"""Build-time gate: no kernel of the library may use scratch memory (a spilled register in a frame loop is a global
memory round trip per frame). Reads the `-Rpass-analysis=kernel-resource-usage` remarks hipcc wrote next to each
object (build/obj/*.remarks), prints anything in them that is not a remark (warnings), and fails when a kernel
reports ScratchSize > 0 — or when one of the kernels whose launch FORM depends on its occupancy has dropped below it: the pool
falls back to a slower form without a word (k_lstm_pipe<32>: BASELINE cfg2's 1024 three-wave workgroups are resident at three
waves per SIMD = 168 registers, and the kernel sits on that line; k_conv_st / k_conv_ms: cfg4's 1024 four-wave workgroups at
four per SIMD = 128 registers). Usage: check_scratch.py <remarks files...>"""
import re
import sys

PAT = re.compile(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                 r"Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)", re.S)


NEED_OCCUPANCY = {                      # mangled-name fragment -> waves per SIMD the launch form needs
    "k_lstm_pipeILi32E": 3,
    "k_conv_stILb": 4,
    "k_conv_msILb1E": 4,                # the fused instantiations
}


def main(files):
    bad, n = [], 0
    for f in files:
        text = open(f).read()
        for line in text.splitlines():
            if "warning:" in line or "error:" in line:
                print(line)
        for name, vgpr, agpr, scratch, occ, sspill, vspill in PAT.findall(text):
            n += 1
            if int(scratch) > 0:
                bad.append(f"{name}: ScratchSize {scratch} B/lane (VGPRs {vgpr}, AGPRs {agpr}, VGPR spills {vspill}, SGPR spills {sspill})")
            for frag, need in NEED_OCCUPANCY.items():
                if frag in name and int(occ) < need:
                    bad.append(f"{name}: occupancy {occ} waves/SIMD (VGPRs {vgpr}), its launch form needs {need}")
    if bad:
        print("kernels using scratch memory or below the occupancy their launch form needs:\n  " + "\n  ".join(bad), file=sys.stderr)
        return 1
    print(f"check_scratch: {n} kernels, none uses scratch memory, the occupancy-bound forms hold theirs")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
