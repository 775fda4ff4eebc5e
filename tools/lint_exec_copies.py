#!/usr/bin/env python3
"""Lint of the COMPILED kernels around the generated assembly (round 5).

The generated loops own v0..v119 (their asm statements clobber them), so whatever per-lane value the compiler wants to keep across
an asm block it copies into one of the few registers left (v120..v127) in front of the block and back behind it.  hipcc 7.2 places
such a copy wherever its live-range splitting likes -- also INSIDE a divergent region (between an s_and_saveexec and the s_or_b64
exec that closes it), where only the active lanes are written; when the value is read again under a wider EXEC mask (a later
flat load, a v_readfirstlane of lane 0), the other lanes come back stale.  That is how the sweep's bound "parked in an LDS word"
read back data-dependent values in round 4: the load's address register had been saved with EXEC = lanes 8..11.

This script compiles a unit to assembly and, kernel by kernel, walks the text with a counter of open saveexec regions; it reports
every `v_mov_b32 v12x, v...` / `v_mov_b32 v..., v12x` (the compiler's carry registers) that executes inside such a region.
The shipped kernels must have none (tests/test_isa_lint.py); -DPROT_SW_THR_LDS=1 -- the round-4 form -- shows what a hit looks like.

    python3 tools/lint_exec_copies.py [-DFLAG ...] [unit.hip]       (default unit: prot_olasave.hip)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "phaserotate.lv2_amd", "csrc")


def compile_asm(unit, defs=()):
    out = "/tmp/_lint_%d.s" % os.getpid()
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-ffast-math",
           "-fno-slp-vectorize", "--cuda-device-only", "-S", os.path.join(CSRC, unit), "-o", out] + list(defs)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stdout[-2000:])
    txt = open(out).read()
    os.unlink(out)
    return txt


def lint(asm_text, carry_lo=120):
    """[(kernel, line number within the kernel, text)] of copies INTO a carry register (v120..) that execute under a narrower EXEC
    mask than the code behind them: in the straight-line code that follows the copy, the next instruction that writes EXEC is an
    `s_or_b64 exec, exec, ...` -- the end of a divergent region the copy sat in.  (Copies out of a carry register, and copies under
    a mask that is narrowed afterwards, are harmless.)"""
    hits = []
    lines = asm_text.splitlines()
    kernel, start, in_asm = None, 0, False
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, start, in_asm = m.group(1), i, False
            continue
        if kernel is None:
            continue
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.startswith("s_endpgm"):
            kernel = None
            continue
        if in_asm:
            continue          # (the generated code manages EXEC itself and is checked by its emulator)
        m = re.match(r"v_mov_b32(_e32)? v(\d+), v(\d+)$", t)
        if not m or int(m.group(2)) < carry_lo:
            continue
        dest = int(m.group(2))
        for j in range(i + 1, min(i + 400, len(lines))):
            u = lines[j].strip()
            if u.startswith(";;#ASMSTART") or re.match(r"s_(c?branch|endpgm|setpc)", u):
                break
            if mentions(u, dest):
                break             # (used or redefined inside the region: a temporary of the region, not a value carried out of it)
            if re.match(r"s_or_b64 exec, exec,", u):
                hits.append((kernel, i - start, t + "   ... " + u))
                break
            if re.search(r"\bexec\b", u) and re.match(r"s_\w+ exec|s_\w+_saveexec", u):
                break
    return hits


def mentions(text, reg):
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        if int(a) <= reg <= int(b):
            return True
    return any(int(x) == reg for x in re.findall(r"\bv(\d+)\b", text))


if __name__ == "__main__":
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    units = [a for a in sys.argv[1:] if a.endswith(".hip")] or ["prot_olasave.hip"]
    bad = 0
    for u in units:
        hits = lint(compile_asm(u, defs))
        sweep = [h for h in hits]
        for k, n, t in sweep:
            name = subprocess.run(["c++filt", k], stdout=subprocess.PIPE, text=True).stdout.strip()
            print("%s: %s  line %d: %s" % (u, name[:70], n, t))
        bad += len(sweep)
        print("%s %s: %d carry-register cop%s under a partial EXEC mask" % (u, " ".join(defs), len(sweep), "y" if len(sweep) == 1 else "ies"))
    sys.exit(1 if bad else 0)
