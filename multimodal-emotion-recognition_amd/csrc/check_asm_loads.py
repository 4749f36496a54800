#!/usr/bin/env python3
"""Build-time guard for kernels that issue global loads from inline asm and wait for them by hand (gemm.hip, Stage16KC).

hipcc treats the destination of an inline-asm load as written when the asm statement ends, so nothing stops it from reading,
copying (`v_mov`, `v_accvgpr_write`), spilling (`scratch_store`) or reusing such a register BEFORE the data has landed - the
copy then holds garbage and the landing load overwrites whatever was allocated to the register meanwhile.  check_spills.py
catches spills; this script catches everything else by reading the ISA (`hipcc -save-temps` .s file):

  * every `global_load_dwordx*` inside an `;;#ASMSTART .. ;;#ASMEND` block puts its destination registers IN FLIGHT;
  * the next inline-asm `s_waitcnt vmcnt(N)` (or a workgroup barrier, or an unconditional branch - the text behind it is
    some other basic block) closes the window: the registers are tracked from the load to the first hand-written wait that
    follows it in fall-through order;
  * any instruction OUTSIDE an asm block that names an in-flight register - as source or destination - is an ERROR, with one
    exception that is only reported as a NOTE: `v_readfirstlane_b32` / `v_readlane_b32` of such a register (hipcc emits one
    behind the last load of a set for a value that is dead on every path - seen in every build of gemm.hip; a scalar read
    cannot corrupt the register, and whatever consumes the scalar shows up in the parity tests);
  * scratch instructions in such a kernel are reported as a NOTE (a scratch store of an in-flight register is an ERROR by
    the rule above).

The scan is linear in program text (it does not follow branches) and covers the window from a load to the first hand-written
wait behind it: the compiler's copies and live-range splits sit right behind the defining statement or in the blocks laid out
after it (that is where the faulty build of round 2 had them: `v_mov_b64` of four just-issued registers in the block in front
of the wait).  Registers that stay in flight ACROSS a counted wait (the second register set of the ring) are not followed
further: without the control flow the text behind the wait mixes the loop's other paths and the consumer waves' code, whose
use of the same register numbers is legitimate.  The check is therefore a necessary condition, not a proof; kernels that
stage through LDS-direct loads (mega.hip) have no destination registers and need none of this.

usage: check_asm_loads.py file.s [kernel-name-substring ...]     (exit 1 on a violation)
"""
import re
import sys

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def scan(lines, name):
    """-> list of violation strings for one kernel body (list of text lines)."""
    bad, notes, inflight, in_asm, uses_asm_loads, scratch = [], [], [], False, False, False
    for no, raw in enumerate(lines, 1):
        line = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else raw.strip()
        if raw.strip().startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if raw.strip().startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.endswith(":") or line.startswith("."):
            continue
        if in_asm:
            m = re.match(r"global_load_dwordx?\d*\s+(v\[\d+:\d+\]|v\d+)", line)
            if m:
                uses_asm_loads = True
                inflight.append(regs_of(m.group(1)))
                continue
            if re.match(r"s_waitcnt\s+.*vmcnt\(\d+\)", line):
                inflight = []                   # end of the window (see the module docstring)
            continue
        if re.match(r"scratch_(load|store)", line):
            scratch = True
        if re.match(r"s_barrier|s_endpgm|s_branch\b|s_setpc", line):     # barrier, or the text behind is another basic block
            inflight = []                       # end of the window (see the module docstring)
            continue
        if inflight:
            live = set().union(*inflight)
            hit = regs_of(line) & live
            if hit:
                msg = f"{name}: line {no}: `{line}` touches v{sorted(hit)} while an inline-asm load into it is in flight"
                (notes if re.match(r"v_read(first)?lane_b32", line) else bad).append(msg)
        m = re.match(r"s_waitcnt\s+.*vmcnt\(0\)", line)
        if m:
            inflight = []                       # a compiler-inserted full drain retires everything as well
    if uses_asm_loads and scratch:
        notes.append(f"{name}: uses scratch memory next to inline-asm loads")
    return bad, notes, uses_asm_loads


def kernels(text):
    """yield (name, body lines) of every function of a .s file"""
    cur, body = None, []
    for line in text.splitlines():
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", line)
        if m and not m.group(1).startswith(".L") and not line.startswith("\t"):
            if cur is not None:
                yield cur, body
            cur, body = m.group(1), []
        elif cur is not None:
            body.append(line)
            if "s_endpgm" in line:
                yield cur, body
                cur, body = None, []
    if cur is not None and body:
        yield cur, body


def main(argv):
    text = open(argv[1], errors="replace").read()
    want = argv[2:]
    violations, remarks, checked = [], [], 0
    for name, body in kernels(text):
        if want and not any(w in name for w in want):
            continue
        bad, notes, uses = scan(body, name)
        if uses:
            checked += 1
        violations += bad
        remarks += notes
    for v in violations[:40]:
        print("check_asm_loads: ERROR", v, file=sys.stderr)
    for v in remarks[:40]:
        print("check_asm_loads: note ", v, file=sys.stderr)
    print(f"check_asm_loads: {checked} kernel(s) with inline-asm loads checked, {len(violations)} error(s), {len(remarks)} note(s)")
    return 1 if violations else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
