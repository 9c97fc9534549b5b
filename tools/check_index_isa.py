#!/usr/bin/env python3
"""Static check of the kernels that load with inline assembly and wait by hand (k_lz4_index in lz4_index.hip, k_zchain4 in
zstd_pipe.hip).

Their ring chunks are inline-assembly global loads whose s_waitcnt is written by hand: the compiler does not know the
destination registers are pending.  Two ways this goes wrong have been seen: (1) register allocation inserts a copy of such a
register in front of the wait (with a `while` loop instead of `do-while`), which reads stale data; (2) once the loop is left
the slots are dead to the compiler and it hands their registers to something else in FRONT of the drain
(`s_waitcnt vmcnt(0)`), where a load still on its way overwrites them (an address, in k_zchain4: a store into the wild).
In k_lz4_index either would only cost speed -- the decoder validates every index entry -- so no test of the output could see
it.  This script compiles the kernels to assembly and requires that, from the loop that holds the assembly loads up to the
drain behind it, a register written by one of those loads is touched by nothing but that load and the ds_write_b128 /
ds_write_b64 that commits it.

The counted waits themselves are checked too: `s_waitcnt vmcnt(N)` in front of a slot's commit is right only if
at least N vector-memory operations are issued between that slot's loads and the wait, once around the loop -- vmcnt counts
in issue order, so with fewer the slot's own loads are among the N that may still be in flight and stale ring bytes are
committed (in k_zchain4: wrong sequences, not merely lost speed).  Operations a forward branch may skip do not count."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pg_cryogen_amd", "csrc")
KERNELS = (("lz4_index.hip", "k_lz4_index"), ("zstd_pipe.hip", "k_zchain4"))


def compile_to_asm(src, extra=()):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-Wno-unused-function", "-Wno-pass-failed", "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, src)] + list(extra)
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(path, kernel):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % kernel, l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end]
    in_asm, asm_lines = False, set()
    for i, l in enumerate(body):
        s = l.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
        elif s.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm:
            asm_lines.add(i)
    asm_load = [i for i in sorted(asm_lines) if body[i].strip().startswith("global_load_dwordx4")]
    if not asm_load:
        return ["no assembly loads found"]
    drains = [i for i in sorted(asm_lines) if body[i].strip() == "s_waitcnt vmcnt(0)"]
    counted = sorted({body[i].strip() for i in asm_lines if re.fullmatch(r"s_waitcnt vmcnt\(\d+\)", body[i].strip())} - {"s_waitcnt vmcnt(0)"})
    problems = []
    if not drains:
        problems.append("no hand-written drain (s_waitcnt vmcnt(0)) behind the loads")
    if len(counted) != 1:
        problems.append("expected one kind of counted wait, found %s" % counted)
    k = 0
    for d in drains:
        grp = [i for i in asm_load[k:] if i < d]
        if not grp:
            continue
        k += len(grp)
        pending = set()
        for i in grp:
            pending |= regs(body[i].split()[1].rstrip(","))
        # the loop: the INNERMOST one that holds all the loads -- a backward branch behind the last load whose target lies at
        # or before the first (an enclosing loop, e.g. a grid-stride one, re-initialises the slots behind the previous drain)
        labels = {body[i].split(":")[0]: i for i in range(0, d) if re.match(r"^\.LBB\d+_\d+:", body[i])}
        first = None
        for i in range(grp[-1], d):
            m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", body[i])
            if m and m.group(1) in labels and labels[m.group(1)] <= grp[0]:
                first = labels[m.group(1)] if first is None else max(first, labels[m.group(1)])
        if first is None:
            problems.append("no loop around the assembly loads at line %d" % (start + grp[0] + 1))
            continue
        # the loop body, then the path from the loop's exit to the drain (unconditional branches followed: a block that lies
        # in between in the text but is only reached by a branch AROUND the loop -- the slots' initial values for a walk
        # that never starts -- is not on that path)
        back = max(i for i in range(grp[-1], d) if re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", body[i]) and
                   labels.get(re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", body[i]).group(1), d) <= grp[0])
        path = list(range(first, back + 1))
        i, guard = back + 1, 0
        while i < d and guard < 100000:
            guard += 1
            path.append(i)
            m = re.match(r"\s*s_branch\s+(\.LBB\d+_\d+)", body[i])
            if m and m.group(1) in labels and labels[m.group(1)] > i:
                i = labels[m.group(1)]
            elif m:
                break   # an unconditional branch backwards or beyond the drain: not the path to it
            else:
                i += 1
        # ---- the counted waits: at least N vector-memory operations between a slot's loads and the wait before its commit ----
        VM = re.compile(r"^(global_|buffer_|scratch_|flat_)")
        loop = list(range(first, back + 1))
        skippable = set()
        for i in loop:
            m = re.match(r"\s*s_cbranch\S*\s+(\.LBB\d+_\d+)", body[i])
            if m and m.group(1) in labels and i < labels[m.group(1)] <= back:
                skippable |= set(range(i + 1, labels[m.group(1)]))
        def is_vm(i):
            return bool(VM.match(body[i].strip())) and i not in skippable
        for ld in [i for i in grp if first <= i <= back]:
            r = regs(body[ld].split()[1].rstrip(","))
            order = [i for i in loop if i > ld] + [i for i in loop if i <= ld] # once around the loop, from behind the load
            commit = next((i for i in order if (body[i].strip().startswith("ds_write_b128") or body[i].strip().startswith("ds_write_b64")) and
                           (regs(re.findall(r"v\[\d+:\d+\]|v\d+", body[i].strip())[-1]) & r)), None)
            if commit is None:
                problems.append("no commit found for the assembly load at line %d" % (start + ld + 1))
                continue
            upto = order[:order.index(commit)]
            waits = [i for i in upto if i in asm_lines and re.fullmatch(r"s_waitcnt vmcnt\(\d+\)", body[i].strip())]
            if not waits:
                problems.append("no counted wait between the assembly load at line %d and its commit" % (start + ld + 1))
                continue
            w = waits[-1]
            n = int(re.search(r"\((\d+)\)", body[w]).group(1))
            younger = sum(1 for i in upto[:upto.index(w)] if is_vm(i))
            if os.environ.get("CRYO_ISA_VERBOSE"):
                print("  %s: load at line %d, vmcnt(%d), %d vector-memory operations behind it" % (kernel, start + ld + 1, n, younger))
            if younger < n:
                problems.append("vmcnt(%d) at line %d: only %d vector-memory operations are issued behind the load at line %d" %
                                (n, start + w + 1, younger, start + ld + 1))
        for i in path:
            s = body[i].strip()
            if not s or s.startswith(";") or s.startswith("."):
                continue
            ops = re.findall(r"v\[\d+:\d+\]|v\d+", s)
            touched = set()
            for t in ops:
                touched |= regs(t)
            if not (touched & pending) or i in grp:
                continue
            if (s.startswith("ds_write_b128") or s.startswith("ds_write_b64")) and regs(ops[-1]) <= pending and not (regs(ops[0]) & pending):
                continue
            problems.append("line %d touches a register of a pending assembly load: %s" % (start + i + 1, s))
    return problems


if __name__ == "__main__":
    bad = 0
    for src, kernel in KERNELS:
        p = compile_to_asm(src, sys.argv[1:])
        pr = check(p, kernel)
        os.unlink(p)
        for x in pr[:20]:
            print("FAIL:", kernel, x)
        print("%s assembly-load check:" % kernel, "FAILED (%d)" % len(pr) if pr else "ok")
        bad += len(pr)
    sys.exit(1 if bad else 0)
