"""ISA lint for the asynchronous window loads of the LDS-tiled kernels (CPU only: hipcc cross-compiles gfx950).

`window_issue6` (kernels.hip) is an inline-asm statement of four buffer loads whose destination registers are ordinary
compiler-allocated variables; the data arrives later, behind the `s_waitcnt vmcnt(N)` of `window_wait6`.  The compiler
does not know that, so nothing between the two statements may make it touch those registers -- a `v_mov` copy placed in
front of the wait reads stale data (seen once in round 3, when a fetch was moved ahead of a block of FMAs).  The parity
tests catch such a build on the GPU; this test catches the commonest form of it here: in the generated ISA, from every
window fetch to the next `s_waitcnt vmcnt` in layout order, no instruction may name one of the fetch's destination
registers."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def regs_of(text):
    """every VGPR index a line of ISA names: v12, v[34:35]"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def lint(src):
    """-> (window fetches found, [(line, instruction)] that touch a fetch's registers before the next vmcnt wait)"""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "kernels_dev.s")
        subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-w",
                               "-I" + os.path.join(ROOT, "s-blas_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                               "--cuda-device-only", "-S", "-o", out, src])
        lines = open(out).read().splitlines()
    sites, bad = 0, []
    i = 0
    while i < len(lines):
        if lines[i].strip() == ";;#ASMSTART":
            j = i + 1
            block = []
            while lines[j].strip() != ";;#ASMEND":
                block.append(lines[j])
                j += 1
            loads = [b for b in block if re.search(r"\bbuffer_load_dword(x2)?\b.*\bidxen\b", b)]
            if len(loads) == 4 and len(block) == 4:                       # a window fetch
                sites += 1
                dest = set()
                for b in loads:
                    dest |= regs_of(b.split(",")[0])                     # first operand = destination
                k = j + 1
                while k < len(lines):
                    t = lines[k].strip()
                    if t.startswith("s_endpgm") or re.search(r"\bs_waitcnt\b.*\bvmcnt\(", t):
                        break
                    if t and not t.startswith((";", ".")) and not t.endswith(":"):
                        if regs_of(t.split(";")[0]) & dest:
                            bad.append((k + 1, t))
                    k += 1
            i = j
        i += 1
    return sites, bad


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_window_fetch_registers_are_left_alone_until_the_wait():
    sites, bad = lint(os.path.join(ROOT, "s-blas_amd", "csrc", "kernels.hip"))
    assert sites >= 20, "found only %d window fetches in the ISA: has the asm changed?" % sites
    assert not bad, "instructions between a window fetch and the next vmcnt wait touch its registers: %s" % bad[:5]
