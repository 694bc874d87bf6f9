#!/usr/bin/env python3
"""Build-time gate: no kernel of the library may use scratch memory (a spilled register in a frame loop is a global
memory round trip per frame). Reads the `-Rpass-analysis=kernel-resource-usage` remarks hipcc wrote next to each
object (build/obj/*.remarks), prints anything in them that is not a remark (warnings), and fails when a kernel
reports ScratchSize > 0. Usage: check_scratch.py <remarks files...>"""
import re
import sys

PAT = re.compile(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                 r"Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)", re.S)


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
    if bad:
        print("kernels using scratch memory:\n  " + "\n  ".join(bad), file=sys.stderr)
        return 1
    print(f"check_scratch: {n} kernels, none uses scratch memory")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
