#!/usr/bin/env python3
"""Static check of k_lz4_index's hand-placed waits (pg_cryogen_amd/csrc/lz4_index.hip).

The index pass loads its ring chunks with inline-assembly global loads whose s_waitcnt is written by hand: the compiler
does not know the destination registers are pending.  If register allocation ever inserted a copy of such a register (it
did, with a `while` loop instead of `do-while`), the copy would read stale data -- and because the decoder validates every
index entry, the result would be slow decoding, never wrong bytes: no test of the output could see it.  So this script
compiles the kernel to assembly and requires that, inside the walk loops, a register written by one of the assembly loads
is touched by nothing but that load and the ds_write_b128 that commits it."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pg_cryogen_amd", "csrc")


def compile_to_asm(extra=()):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-Wno-unused-function", "-Wno-pass-failed", "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, "lz4_index.hip")] + list(extra)
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(path):
    lines = open(path).read().splitlines()
    # kernel body
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*k_lz4_index\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end]
    asm_load, in_asm = [], False
    for i, l in enumerate(body):
        s = l.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
        elif s.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm and s.startswith("global_load_dwordx4"):
            asm_load.append(i)
    if not asm_load:
        return ["no assembly loads found"]
    pending = set()
    for i in asm_load:
        pending |= regs(body[i].split()[1].rstrip(","))
    problems = []
    # walk regions: from the first assembly load's loop to the drain that follows it (s_waitcnt vmcnt(0) inside an ASM block)
    drains = [i for i, l in enumerate(body) if l.strip() == "s_waitcnt vmcnt(0)" and body[i - 1].strip().startswith(";;#ASMSTART")]
    waits = [i for i, l in enumerate(body) if re.fullmatch(r"s_waitcnt vmcnt\(\d+\)", l.strip()) and body[i - 1].strip().startswith(";;#ASMSTART")]
    if not drains:
        problems.append("no hand-written drain (s_waitcnt vmcnt(0)) behind the walk")
    counted = [body[i].strip() for i in waits if body[i].strip() != "s_waitcnt vmcnt(0)"]
    if not counted or len(set(counted)) != 1:
        problems.append("expected one kind of counted wait, found %s" % sorted(set(counted)))
    for lo_hi in _regions(asm_load, drains):
        lo, hi = lo_hi
        # the walk loop: the outermost backward branch between the loads and the drain, and its target label
        labels = {body[i].split(":")[0]: i for i in range(0, hi) if re.match(r"^\.LBB\d+_\d+:", body[i])}
        first, last = None, None
        for i in range(lo, hi):
            m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", body[i])
            if m and m.group(1) in labels and labels[m.group(1)] <= lo:
                first = labels[m.group(1)] if first is None else min(first, labels[m.group(1)])
                last = i
        if first is None:
            problems.append("no loop around the assembly loads at line %d" % (start + lo + 1))
            continue
        hi = last + 1
        for i in range(first, hi):
            s = body[i].strip()
            if not s or s.startswith(";") or s.startswith("."):
                continue
            ops = re.findall(r"v\[\d+:\d+\]|v\d+", s)
            touched = set()
            for t in ops:
                touched |= regs(t)
            if not (touched & pending):
                continue
            if i in asm_load:
                continue
            if s.startswith("ds_write_b128") and regs(ops[-1]) <= pending and not (regs(ops[0]) & pending):
                continue
            problems.append("line %d touches a register of a pending assembly load: %s" % (start + i + 1, s))
    return problems


def _regions(loads, drains):
    out, k = [], 0
    for d in drains:
        grp = [i for i in loads[k:] if i < d]
        if grp:
            out.append((grp[0], d))
            k += len(grp)
    return out


if __name__ == "__main__":
    p = compile_to_asm(sys.argv[1:])
    pr = check(p)
    os.unlink(p)
    for x in pr[:20]:
        print("FAIL:", x)
    print("k_lz4_index assembly-load check:", "FAILED (%d)" % len(pr) if pr else "ok")
    sys.exit(1 if pr else 0)
