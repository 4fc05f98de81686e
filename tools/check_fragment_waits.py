#!/usr/bin/env python3
"""Lint for the hand-scheduled kernels: a register written by a ds_read must not be READ before an s_waitcnt lgkmcnt that covers
that read.  The kernels issue their LDS fragment reads from inline asm and count the waits by hand; the compiler believes such a
register is defined as soon as the asm has been issued, so a copy it places at a merge of two definitions (loop heads, tile
boundaries, conditional requests) copies a fragment that has not arrived -- the fault class of DESIGN.md section 4.
Linear scan per kernel in layout order, restarted behind every unconditional branch (LDS reads return in order, lgkmcnt(n) leaves the n youngest pending; scalar loads and
LDS writes also count and are tracked as anonymous entries).  Reports the first offenders per kernel.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only csrc/pgemm.hip -o /tmp/pg.s && python tools/check_fragment_waits.py /tmp/pg.s [filter]"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|v\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    name, pending, bad, total_bad = None, [], [], 0      # pending: list of (set of dest registers | None, line number)

    def close():
        nonlocal name, total_bad
        if name is not None and bad and flt in name:
            print(f"{name}: {len(bad)} read(s) of a pending LDS destination")
            for b in bad[:6]:
                print("   ", b)
            total_bad += len(bad)
        name = None

    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        if t.startswith("_Z") and ":" in t and "kernel" in t.split(":")[0]:
            close()
            name, pending, bad = t.split(":")[0], [], []
            continue
        # a kernel's code ends at its function-end label (code laid out behind an early-exit s_endpgm still belongs to it)
        if t.startswith((".Lfunc_end", ".end_amdhsa_kernel")):
            close()
            continue
        if name is None or not t or t.startswith((";", ".", "//")):
            continue
        if t.startswith("s_endpgm"):               # like an unconditional branch: what follows is reached from elsewhere
            pending = []
            continue
        op = t.split()[0]
        body = t[len(op):].split(";")[0]
        if op in ("s_branch", "s_setpc_b64"):      # the code behind an unconditional jump is reached from elsewhere: unknown
            pending = []                           # state, taken as clean (a layout-order scan would report reads of the OTHER path)
            continue
        if op.startswith("s_waitcnt"):
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if n < len(pending) else pending
                if n == 0:
                    pending = []
            continue
        ops = [x.strip() for x in body.split(",")]
        if op.startswith("ds_read") or op.startswith("ds_load"):
            srcs = set().union(*[regs(x) for x in ops[1:]]) if len(ops) > 1 else set()
            for dst, l0 in pending:
                if dst and dst & srcs:
                    bad.append(f"line {ln}: {t}   (address register pending since line {l0})")
            pending.append((regs(ops[0]), ln))
            continue
        if op.startswith(("ds_write", "ds_store", "ds_bpermute", "ds_swizzle", "s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
            pending.append((None, ln))       # counts in lgkmcnt, no vector destination to guard (bpermute: treated below)
            continue
        # any other instruction: every vector register it names (sources AND destinations: overwriting a pending one is as bad)
        used = set().union(*[regs(x) for x in ops]) if ops else set()
        for dst, l0 in pending:
            if dst and dst & used:
                bad.append(f"line {ln}: {t}   (LDS read issued at line {l0} not waited for)")
                break
    close()
    print("checked", path, "->", "OK" if total_bad == 0 else f"{total_bad} offending instruction(s)")
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
