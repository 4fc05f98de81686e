#!/usr/bin/env python3
"""Scratch (register spills to memory) of the kernels named on the command line, from hipcc's -Rpass-analysis=kernel-resource-usage
remarks (tools/lint_kernels.sh writes them next to the ISA listings).  The plane GEMM kernels must have none: their epilogues
spilled 68 - 352 bytes per lane in round 4 (the reloads sat in every epilogue form, paid once per tile).
    python tools/check_scratch.py <remarks file>... -- <kernel name substring>..."""
import re
import sys


def main():
    args = sys.argv[1:]
    cut = args.index("--")
    files, names = args[:cut], args[cut + 1:]
    bad, seen = [], 0
    for path in files:
        name = None
        for line in open(path, errors="replace"):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]): (\d+)", line)
            if m and name and any(n in name for n in names):
                if m.group(1) == "VGPRs":
                    vg = int(m.group(2))
                else:
                    seen += 1
                    print(f"{name[:110]:110s} VGPRs {vg:3d}  scratch {m.group(2)} B")
                    if int(m.group(2)):
                        bad.append(name)
    print("scratch check:", seen, "kernels ->", "OK" if seen and not bad else f"{len(bad)} kernel(s) with scratch" if bad else "no kernel matched")
    return 1 if bad or not seen else 0


if __name__ == "__main__":
    sys.exit(main())
