#!/usr/bin/env python3
"""Looks for matrix-instruction hazards that hipcc cannot pad because one side sits in inline asm.

    python devtest/mfma_asm_scan.py dev.s [kernel-name-substring ...]

On gfx90a / gfx940 / gfx950 the hardware does not interlock a vector-ALU instruction against the
VGPRs of a matrix instruction (MFMA) still in flight: the compiler's hazard recognizer inserts the
wait states (LLVM GCNHazardRecognizer::checkMAIVALUHazards):
  * RAW  a VALU read of the MFMA's destination      (passes + 2 ... + 3 wait states),
  * WAW  a VALU write of the MFMA's destination     (the same),
  * WAR  a VALU write of the MFMA's SrcC registers  (up to `passes` wait states; only matters when
         the MFMA's destination is NOT its SrcC, i.e. the SrcC registers are dead after the issue).
It looks at machine instructions, and an inline-asm statement is opaque to it: nothing is padded
between an MFMA and an asm statement, whatever the statement contains.  This walk reports every
instruction between `;;#ASMSTART` / `;;#ASMEND` that touches the registers of an MFMA issued fewer
than WINDOW wait states earlier (straight-line; `s_nop N` counts N + 1, every other instruction 1).
"""
import re
import sys

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
WINDOW = 20


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def operands(ins):
    parts = ins.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    return parts[0], [p.strip() for p in parts[1].split(",")]


def scan(path, wanted):
    text = open(path).read()
    hits = 0
    for m in re.finditer(r"^(_Z[\w]+):[^\n]*$", text, re.M):
        name = m.group(1)
        if "ycnr" not in name or (wanted and not any(w in name for w in wanted)):
            continue
        end = text.find("s_endpgm", m.end())
        body = text[m.end():end].split("\n")
        recent = []  # (age, lineno, ins, vdst set, srcC set)
        inasm = False
        khits = 0
        for ln, raw in enumerate(body):
            s = raw.strip()
            if s.startswith(";;#ASMSTART"):
                inasm = True
                continue
            if s.startswith(";;#ASMEND"):
                inasm = False
                continue
            if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
                continue
            ins = s.split(";")[0].strip()
            if not ins:
                continue
            op, ops = operands(ins)
            step = 1
            if op == "s_nop":
                step = int(ops[0]) + 1 if ops else 1
            if inasm and op.startswith("v_") and recent:
                if "swap" in op:
                    wr = regs(ops[0]) | regs(ops[1])
                    rd = wr
                else:
                    wr = regs(ops[0]) if ops else set()
                    rd = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
                    if "fmac" in op or "_mac_" in op:
                        rd |= wr
                for age, mln, mins, vdst, srcc in recent:
                    kinds = []
                    if rd & vdst:
                        kinds.append("RAW")
                    if wr & vdst:
                        kinds.append("WAW")
                    if wr & (srcc - vdst):
                        kinds.append("WAR")
                    if kinds:
                        print("%s\n  line +%d  %s\n  %d wait states behind (+%d)  %s   [%s]" %
                              (name[:110], ln, ins, age, mln, mins, ",".join(kinds)))
                        hits += 1
                        khits += 1
            recent = [(a + step, l, i, d, c) for (a, l, i, d, c) in recent if a + step < WINDOW]
            if op.startswith("v_mfma") or op.startswith("v_smfma"):
                vdst = regs(ops[0])
                srcc = regs(ops[3]) if len(ops) > 3 else set()
                recent.append((0, ln, ins, vdst, srcc))
            elif op in ("s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_execz",
                        "s_cbranch_execnz", "s_branch", "s_barrier"):
                pass  # straight-line walk: a branch neither resets nor extends the window
    return hits


if __name__ == "__main__":
    n = scan(sys.argv[1], sys.argv[2:])
    print("mfma_asm_scan: %d candidate(s)" % n)
    sys.exit(0)
