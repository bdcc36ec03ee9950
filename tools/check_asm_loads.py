#!/usr/bin/env python3
"""Static check of kernels whose global loads and vmcnt waits are hand-written inline assembly (wg3.hip's loader waves).

hipcc does not count a load issued from inline assembly, so NOTHING but the hand-written `s_waitcnt vmcnt(N)` orders a use of the
loaded registers behind the arrival of the data.  The C++ expresses that as data flow (the wait takes the registers as read-write
operands); this script checks the result in the ISA: between an inline-assembly `global_load_dwordx4 v[a:b]` and the hand-written
wait that retires it, no other instruction may mention v[a..b] (a copy, a spill or a use would read registers the load has not
written yet).  Model: loads retire in order; a wait vmcnt(N) retires all but the N newest.  Loop bodies are scanned twice so that
loads carried around the back edge are seen by the code at the top of the loop.

    python tools/check_asm_loads.py <file.s | file.hip> [kernel-name-substring]      exit status 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile


def device_asm(path):
    if path.endswith(".s"):
        return open(path).read()
    out = tempfile.mktemp(suffix=".s")
    inc = os.path.dirname(os.path.abspath(path))
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-S", "--cuda-device-only",
                    "-I", inc, path, "-o", out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def regs_of(line):
    """All VGPR numbers an instruction line mentions."""
    body = line.split(";")[0]
    found = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", body):
        found.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", body):
        found.add(int(a))
    return found


def check_kernel(name, lines):
    """lines: the kernel's body.  Returns (number of asm loads, list of violations)."""
    in_asm = False
    events = []  # (kind, payload, lineno, text): kind in load / wait / insn / label / branch
    for no, raw in enumerate(lines):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", t)
            if m:
                events.append(("label", m.group(1), no, t))
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            events.append(("label", m.group(1), no, t))
            continue
        if in_asm and t.startswith("global_load_dwordx4"):
            dst = re.match(r"global_load_dwordx4\s+v\[(\d+):(\d+)\]", t)
            # the address registers are read at issue (before the destination is written: they may overlap it): a use of whatever
            # is pending in them
            addr = regs_of(t.split(",", 1)[1])
            events.append(("insn", addr, no, t))
            events.append(("load", set(range(int(dst.group(1)), int(dst.group(2)) + 1)), no, t))
            continue
        if in_asm and t.startswith("global_store_dwordx4"):
            # a hand-written store: vmcnt counts it like a load (gfx9-family: one counter, retired in issue order); it reads its operands
            # at issue and defines nothing
            events.append(("insn", regs_of(t.split(None, 1)[1]), no, t))
            events.append(("load", set(), no, t))
            continue
        if t.startswith("global_load_lds_dwordx4") or (t.startswith("global_load_lds") and "dword" in t):
            # an LDS-DMA request (the compiler's builtin or hand-written): it takes a slot of the in-order vmcnt queue like any load and
            # writes no register (pig.hip's K-deep form interleaves the weight DMA with its hand-written row loads: the counted waits
            # only mean what they say if the checker's queue holds both)
            events.append(("insn", regs_of(t.split(None, 1)[1]), no, t))
            events.append(("load", set(), no, t))
            continue
        if in_asm and t.startswith("s_waitcnt") and "vmcnt" in t:
            n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
            events.append(("wait", n, no, t))
            continue
        if re.match(r"^s_c?branch", t):
            tgt = t.split()[-1]
            events.append(("branch", tgt, no, t))
            continue
        events.append(("insn", regs_of(t), no, t))
    nloads = sum(1 for e in events if e[0] == "load")
    if nloads == 0:
        return 0, []
    # basic blocks: split at labels and behind branches; a forward data-flow over the control-flow graph carries the queue of
    # loads in flight (issue order) along every path - path-sensitive, states memoised per block
    blocks, cur = [], []
    for e in events:
        if e[0] == "label" and cur:
            blocks.append(cur)
            cur = []
        cur.append(e)
        if e[0] == "branch":
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    first_label = {}
    for bi, b in enumerate(blocks):
        if b[0][0] == "label":
            first_label[b[0][1]] = bi
    succ = []
    for bi, b in enumerate(blocks):
        out = []
        last = b[-1]
        if last[0] == "branch":
            if last[1] in first_label:
                out.append(first_label[last[1]])
            if not last[3].startswith("s_branch") and bi + 1 < len(blocks):
                out.append(bi + 1)   # conditional: falls through as well
        elif bi + 1 < len(blocks):
            out.append(bi + 1)
        succ.append(out)
    violations = set()
    seen = set()
    work = [(0, ())]
    while work:
        bi, pending = work.pop()
        if (bi, pending) in seen:
            continue
        seen.add((bi, pending))
        if len(seen) > 200000:
            violations.add(f"{name}: state space too large for the checker")
            break
        q = list(pending)
        for kind, payload, no, text in blocks[bi]:
            if kind == "load":
                q.append((frozenset(payload), no, text))
            elif kind == "wait":
                while len(q) > payload:
                    q.pop(0)
            elif kind == "insn":
                for regs, lno, ltext in q:
                    hit = regs & payload
                    if hit:
                        violations.add(f"{name}: line {no}: `{text}` touches v{sorted(hit)} loaded at line {lno} (`{ltext}`) before its wait")
        for nb in succ[bi]:
            work.append((nb, tuple(q)))
    return nloads, sorted(violations)


def scalar_offset_loads(name, lines):
    """Scalar loads whose address is  base + SGPR offset (+ immediate).  hipcc (ROCm 7.2) built one for `args.ptr[phase]` - a kernel
    argument array indexed by a run-time scalar - as base = kernarg + phase, offset = 7 * phase + 0x278: a base that is not dword-aligned,
    whose low two bits the scalar memory unit ignores, so phases 1-3 of hf.hip read a wrong pointer (HSA_STATUS_ERROR_MEMORY_APERTURE_
    VIOLATION at bring-up).  The checked kernels select such values from constant-index copies instead; any register-offset scalar load
    in them is reported."""
    out = []
    for no, text in enumerate(lines, 1):
        t = text.split(";")[0].strip()
        if re.match(r"s_(buffer_)?load_dword\w*\s+s\S+,\s*s\[\d+:\d+\],\s*s\d+", t):
            out.append(f"{name}: line {no}: `{t}` is a scalar load with a register offset (run-time index into a kernel argument?)")
    return out


def main():
    text = device_asm(sys.argv[1])
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    kernels = re.split(r"\n(?=_Z[\w]+:\s*;\s*@)", text)
    total, bad = 0, []
    for k in kernels:
        m = re.match(r"(_Z[\w]+):", k)
        if not m or want not in m.group(1):
            continue
        body = k.split("s_endpgm")[0].splitlines()
        n, v = check_kernel(m.group(1), body)
        v += scalar_offset_loads(m.group(1), body)
        total += n
        bad += v
        if n:
            print(f"{m.group(1)}: {n} inline-assembly loads, {len(v)} violations")
    for v in bad:
        print("VIOLATION", v)
    if total == 0:
        print("no inline-assembly global loads found")
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
