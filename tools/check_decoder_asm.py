#!/usr/bin/env python3
"""Build-time audit of the decoder's hand-issued loads (zig-lz4_amd/csrc/zlz4_decompress.hip).

k_decompress_safe issues three kinds of loads from asm statements with plain register outputs and waits for them in
separate asm `s_waitcnt` statements (the compiler's own vmcnt bookkeeping would drain the queue at the first use):
  * the window load (`global_load_dword`) and the match loads (`global_load_dwordx4`): awaited by the batch's
    `s_waitcnt vmcnt(1)` -- everything but the youngest operation -- or by a `vmcnt(0)`;
  * the far-ahead touch (`global_load_ubyte`), issued as the LAST vector-memory operation of a batch: it is the one
    operation a `vmcnt(1)` leaves in flight, so it is only covered by a `vmcnt(0)`, or by a `vmcnt(1)` that comes after
    the NEXT touch has been issued.
hipcc does not know these registers are in flight: a copy, spill or reuse of one of them between issue and wait would
read stale data silently (cdna_hip_programming.md section 5.7 item 1).  This script compiles the file to assembly and
checks, for every such load, that no instruction outside the asm statements names a destination register on any
control-flow path from the load to the wait that covers it.  Run by `make check-asm` and tests/test_capi.py; re-run it
-- and the decoder parity tests on the GPU -- after any toolchain or flag change.
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
        load_re = re.compile(r"^global_load_(dword|dwordx4|ubyte)\s+(v\d+|v\[\d+:\d+\])")
        for i in range(k0, k1 + 1):
            t = lines[i].strip()
            m0 = load_re.match(t) if asm_region[i] else None
            if not m0:
                continue
            kind = m0.group(1)
            dest = regs_in(m0.group(2))
            checked += 1
            # state: (line, a newer touch has been issued since this load)
            seen, stack, waits = set(), [(i + 1, False)], 0
            while stack:
                j, newer = stack.pop()
                while j <= k1:
                    if (j, newer) in seen:
                        break
                    seen.add((j, newer))
                    t = lines[j].strip().split(";")[0].strip() if not lines[j].strip().startswith(";;") else ""
                    if not t or t.startswith((".", "//")) or t.endswith(":"):
                        j += 1
                        continue
                    if asm_region[j] and t.startswith("s_waitcnt"):
                        if "vmcnt(0)" in t or ("vmcnt(1)" in t and (kind != "ubyte" or newer)):
                            waits += 1
                            break                               # this path is covered
                        j += 1
                        continue
                    if t.startswith("s_endpgm"):
                        problems.append("the load at line %d reaches s_endpgm without a covering wait" % (i + 1))
                        break
                    if asm_region[j] and load_re.match(t):
                        if t.startswith("global_load_ubyte"):
                            newer = True                        # (writes the touch register again: still asm, still fine)
                        elif dest & regs_in(load_re.match(t).group(2)):
                            problems.append("v%s (in flight since line %d) is loaded again at line %d before its wait" % (sorted(dest), i + 1, j + 1))
                        j += 1
                        continue
                    if dest & regs_in(t):
                        problems.append("v%s (in flight since line %d) is touched at line %d: %s" % (sorted(dest), i + 1, j + 1, t))
                    m = re.match(r"^(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)", t)
                    if m:
                        tgt = label_at.get(m.group(2))
                        if tgt is not None:
                            stack.append((tgt, newer))
                        if m.group(1) == "s_branch":
                            break
                    j += 1
            if waits == 0:
                problems.append("no covering asm s_waitcnt reachable from the load at line %d" % (i + 1))
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
