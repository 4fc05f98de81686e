"""The hand-scheduled kernels issue their LDS fragment reads from inline asm and count the waits by hand; the compiler believes a
destination register is defined as soon as the asm has been issued, so a copy or a spill it places between the read and the wait
moves data that has not arrived (the intermittent faults of round 4).  tools/lint_kernels.sh compiles every
such translation unit to ISA (no GPU needed) and tools/check_fragment_waits.py scans it for exactly that."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_kernel_reads_an_lds_destination_before_its_wait(tmp_path):
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "lint_kernels.sh")], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, TMPDIR=str(tmp_path)))
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("checked")]
    assert len(lines) == 12 and all(ln.endswith("-> OK") for ln in lines), r.stdout[-3000:] + r.stderr[-1000:]
    # every pgemm_nt / pgemm_tn instantiation (the epilogues spilled 68 - 352 B / lane in round 4) and every attention kernel of the
    # four families (pattn_fwd<2> and mattn_fwd_long<16> spilled 20 B in round 5): no scratch (tools/check_scratch.py)
    scratch = [ln for ln in r.stdout.splitlines() if ln.startswith("scratch check:")]
    assert len(scratch) == 3 and all(ln.endswith("-> OK") for ln in scratch), r.stdout[-3000:]
    assert int(scratch[0].split()[2]) >= 11 and int(scratch[1].split()[2]) >= 104 and int(scratch[2].split()[2]) >= 7, scratch          # kernels actually matched
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
