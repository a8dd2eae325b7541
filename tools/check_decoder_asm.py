#!/usr/bin/env python3
"""Build-time audit of the decoder's hand-issued window prefetch (zig-lz4_amd/csrc/zlz4_decompress.hip).

k_decompress_safe issues `global_load_dword` from an asm statement with a plain "=v" output and waits for it in a
separate asm `s_waitcnt vmcnt(0)` (the compiler's own vmcnt bookkeeping would drain the queue earlier).  hipcc does not
know the register is in flight: a copy, spill or reuse of that VGPR between the two statements would read stale data
silently (cdna_hip_programming.md section 5.7 item 1).  This script compiles the file to assembly and checks, for every
such load, that no instruction on any control-flow path from it to the next asm `s_waitcnt vmcnt(0)` names the destination
register.  Run by `make check-asm` and tests/test_capi.py; re-run it -- and
the decoder parity tests on the GPU -- after any toolchain or flag change.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "zig-lz4_amd", "csrc", "zlz4_decompress.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def regs_in(line):
    out = set()
    for m in re.finditer(r"\bv(\d+)\b", line):
        out.add(int(m.group(1)))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def audit(asm_text):
    """Walk the control-flow graph from every hand-issued load to the asm waits that can follow it."""
    lines = asm_text.splitlines()
    # kernel extents
    kernels, start = [], None
    for i, ln in enumerate(lines):
        if re.match(r"^_ZN4zlz417k_decompress_safe", ln):
            start = i
        elif start is not None and "s_endpgm" in ln:
            kernels.append((start, i))
            start = None
    problems, checked = [], 0
    for k0, k1 in kernels:
        label_at = {}
        for i in range(k0, k1 + 1):
            m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
            if m:
                label_at[m.group(1)] = i
        asm_region = [False] * (k1 + 2)
        inside = False
        for i in range(k0, k1 + 1):
            if ";;#ASMSTART" in lines[i]:
                inside = True
            asm_region[i] = inside
            if ";;#ASMEND" in lines[i]:
                inside = False
        for i in range(k0, k1 + 1):
            t = lines[i].strip()
            if not (asm_region[i] and t.startswith("global_load_dword ")):
                continue
            dest = int(re.search(r"global_load_dword\s+v(\d+)", t).group(1))
            checked += 1
            seen, stack, waits = set(), [i + 1], 0
            while stack:
                j = stack.pop()
                while j <= k1:
                    if j in seen:
                        break
                    seen.add(j)
                    t = lines[j].strip().split(";")[0].strip() if not lines[j].strip().startswith(";;") else ""
                    if not t or t.startswith((".", "//")) or t.endswith(":"):
                        j += 1
                        continue
                    if asm_region[j] and t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                        waits += 1
                        break                                   # this path is covered
                    if t.startswith("s_endpgm"):
                        break
                    if not (asm_region[j] and t.startswith("global_load_dword ")) and dest in regs_in(t):
                        problems.append("v%d (in flight since line %d) is touched at line %d: %s" % (dest, i + 1, j + 1, t))
                    m = re.match(r"^(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)", t)
                    if m:
                        tgt = label_at.get(m.group(2))
                        if tgt is not None:
                            stack.append(tgt)
                        if m.group(1) == "s_branch":
                            break
                    j += 1
            if waits == 0:
                problems.append("no asm s_waitcnt vmcnt(0) reachable from the load at line %d" % (i + 1))
    return checked, problems


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "d.s")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S",
                               "--cuda-device-only", SRC, "-o", out], stderr=subprocess.DEVNULL)
        checked, problems = audit(open(out).read())
    print("decoder asm audit: %d hand-issued loads checked, %d problems" % (checked, len(problems)))
    for p in problems:
        print("  " + p)
    return 1 if problems or checked == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
